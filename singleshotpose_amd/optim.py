"""SGD on one fused HIP launch - the optimizer side of the training step.

The reference trains with `optim.SGD(model.parameters(), lr=learning_rate/batch_size, momentum=momentum, dampening=0,
weight_decay=decay*batch_size)` (train.py:388), `optimizer.zero_grad()` / `optimizer.step()` per batch
(train.py:89,106) and rewrites `param_group['lr']` every batch (train.py:44-45).  `SGD` below keeps that constructor,
`param_groups`, `zero_grad`, `state_dict` (it is a torch.optim.Optimizer) and the update rule of torch.optim.SGD, but
runs the whole step as ONE ssp_sgd_step launch:

* Plan.backward (engine.py) already returns every parameter gradient as a view of one flat fp32 buffer, in reverse
  layer order (the buffer the RCCL all-reduce works on).  At the first step the optimizer adopts exactly that layout:
  it moves the parameters into one flat buffer (each `p.data` becomes a view of it - the module tree is unchanged)
  and allocates one flat momentum buffer, so parameter i, its gradient and its momentum sit at the same offset.
* A step whose gradients are such views (the normal case) is then a single pass over 3 x 202 MB; gradients that are
  not (accumulated over several backwards, clipped copies, ...) are handled per parameter with the same kernel.

No CPU path: parameters must be on the GPU.
"""
import torch

from . import _lib
from .engine import weights_changed


def _dense(t):
    """Contiguous, or channels-last (conv filters as Darknet keeps them): numel consecutive floats either way."""
    return t.is_contiguous() or (t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous())


def _same_layout(a, b):
    """Same shape and the same strides on every dimension that has more than one element."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: %s" % lr)
        if momentum < 0.0:
            raise ValueError("Invalid momentum value: %s" % momentum)
        if weight_decay < 0.0:
            raise ValueError("Invalid weight_decay value: %s" % weight_decay)
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super(SGD, self).__init__(params, defaults)
        self._flat_p = None      # flat parameter buffer (views handed to the modules)
        self._flat_m = None      # flat momentum buffer, same layout
        self._layout = None      # [(param, offset, numel)]
        self._first = True
        self.fused_steps = 0     # steps done as one launch (introspection / tests)

    # ---- flat layout ---------------------------------------------------------------------------------------------
    def _all_params(self):
        return [p for g in self.param_groups for p in g['params']]

    @staticmethod
    def _flat_grads(params):
        """(storage base pointer, [(param, offset)], total) when every .grad is a float32 view of ONE storage with
        non-overlapping 16-byte aligned ranges, else None."""
        base = None
        items = []
        for p in params:
            g = p.grad
            if g is None or g.dtype != torch.float32 or not g.is_cuda:
                return None
            # dense in memory with the parameter's own strides (contiguous, or channels-last conv filters)
            if not (_dense(p) and _same_layout(g, p)):
                return None
            st = g.untyped_storage()
            if base is None:
                base = st
            elif st.data_ptr() != base.data_ptr():
                return None
            off = g.storage_offset()
            if off % 4:
                return None
            items.append((p, off))
        if base is None:
            return None
        order = sorted(items, key=lambda t: t[1])
        for (pa, oa), (pb, ob) in zip(order[:-1], order[1:]):
            if oa + pa.numel() > ob:
                return None
        total = base.nbytes() // 4
        return base, items, total

    def _adopt_layout(self, items, total, device):
        flat_p = torch.zeros(total, dtype=torch.float32, device=device)
        flat_m = torch.zeros(total, dtype=torch.float32, device=device)
        layout = []
        for p, off in items:
            n = p.numel()
            # views with the parameter's own strides (channels-last filters stay channels-last), element for element
            # at the offsets of the gradient views
            pv = torch.as_strided(flat_p, p.shape, p.stride(), off)
            mv = torch.as_strided(flat_m, p.shape, p.stride(), off)
            pv.copy_(p.data)
            old = self.state.get(p, {}).get('momentum_buffer')
            if old is not None:
                mv.copy_(old)
                self._first = False      # momentum carried over (per-parameter steps or load_state_dict): not a first step
            p.data = pv
            self.state[p]['momentum_buffer'] = mv
            layout.append((p, off, n))
        self._flat_p, self._flat_m, self._layout = flat_p, flat_m, layout

    def _layout_valid(self, items):
        if self._layout is None or len(items) != len(self._layout):
            return False
        base = self._flat_p.data_ptr()
        mbase = self._flat_m.data_ptr()
        for (p, off), (q, qoff, n) in zip(items, self._layout):
            if p is not q or off != qoff or p.data_ptr() != base + 4 * off:
                return False      # model.cuda()/load_state_dict replaced a tensor, or the plan's layout changed
            m = self.state.get(p, {}).get('momentum_buffer')
            if m is None or m.data_ptr() != mbase + 4 * off:
                return False      # optimizer.load_state_dict() swapped the momentum tensors: re-adopt (copies them in)
        return True

    def _uniform_hyper(self):
        g0 = self.param_groups[0]
        keys = ('lr', 'momentum', 'dampening', 'weight_decay', 'nesterov')
        for g in self.param_groups[1:]:
            if any(g[k] != g0[k] for k in keys):
                return None
        return g0

    # ---- step ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        params = [p for p in self._all_params() if p.grad is not None]
        if not params:
            return loss
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("singleshotpose_amd.optim.SGD runs on the MI355X HIP kernel only: parameter on %s" % p.device)
        st = torch.cuda.current_stream(params[0].device).cuda_stream
        hyper = self._uniform_hyper()
        flat = self._flat_grads(params) if (hyper is not None and len(params) == len(self._all_params())) else None
        if flat is not None:
            base, items, total = flat
            if not self._layout_valid(items):
                self._adopt_layout(items, total, params[0].device)
                # momentum carried over from per-parameter steps keeps `first` as it was
            g0 = params[0].grad
            gbase = g0.data_ptr() - 4 * g0.storage_offset()
            first = 1 if (self._first and hyper['momentum'] != 0) else 0
            _lib.call('ssp_sgd_step', self._flat_p.data_ptr(), gbase, self._flat_m.data_ptr(), total,
                      float(hyper['lr']), float(hyper['momentum']), float(hyper['dampening']),
                      float(hyper['weight_decay']), 1 if hyper['nesterov'] else 0, first, st)
            self._first = False
            self.fused_steps += 1
            weights_changed()      # raw-pointer writes do not bump tensor versions: invalidate packed-filter caches
            return loss
        # per-parameter launches: mixed hyper-parameters, missing or non-flat gradients
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad
                if not _dense(p.data):
                    raise RuntimeError("SGD: parameter is neither contiguous nor channels-last")
                if g.dtype != torch.float32 or not _same_layout(g, p):
                    g = torch.empty_like(p.data).copy_(g)      # same memory order as the parameter
                state = self.state[p]
                buf = state.get('momentum_buffer')
                first = 0
                if group['momentum'] != 0 and buf is None:
                    buf = torch.zeros_like(p.data)
                    state['momentum_buffer'] = buf
                    first = 1
                if (p.data_ptr() | g.data_ptr() | (buf.data_ptr() if buf is not None else 0)) & 15:
                    raise RuntimeError("SGD: parameter / gradient storage is not 16-byte aligned")
                _lib.call('ssp_sgd_step', p.data_ptr(), g.data_ptr(), buf.data_ptr() if buf is not None else None,
                          p.numel(), float(group['lr']), float(group['momentum']), float(group['dampening']),
                          float(group['weight_decay']), 1 if group['nesterov'] else 0, first, st)
        self._first = False
        weights_changed()
        return loss
