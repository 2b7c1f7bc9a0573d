"""Execution plan for a Darknet cfg on the HIP kernels (libssp_hip.so).

Walks the block list the way Darknet.forward does (/root/reference/darknet.py:82-130) but emits kernel launches
through the C ABI instead of torch.nn modules.  Activations live in NHWC fp32 buffers owned by torch (allocated once
per input shape and reused every step); every launch goes to torch's current HIP stream.

Forward per conv block (darknet.py:145-167):
    repack filters -> ssp_conv_fwd (MFMA implicit GEMM, BN partial statistics in the epilogue)
    -> ssp_bn_fwd_finalize (train) | ssp_bn_eval_prepare (eval) -> ssp_bn_act_fwd (BN + leaky [+ fused 2x2 max-pool])
A max-pool block is fused into the preceding conv block when nothing else consumes the un-pooled map.
route = alias or ssp_copy_channels into a concat buffer; reorg = ssp_reorg.

Backward mirrors it in reverse: ssp_bn_act_bwd (in place over the raw conv output) -> ssp_conv_wgrad ->
ssp_unpack_grad, and ssp_conv_dgrad into the producer's gradient buffer (accumulating when a map has two consumers).
"""
import os
import weakref

import torch

from . import _lib
from .cfg import layer_shapes, resolve_layers

WINO = 9000000       # plan codes WINO + tile_rows*100 + 10 + ring_slots: Winograd F(2x2, 3x3) evaluation (csrc/conv_wino.hip)
WINO4 = 8000000      # ... and WINO4 + the same: F(4x4, 3x3)
WGRAD_FUSED = 12     # `tile` value of ssp_conv_wgrad_wino_t: F(2x2) filter gradient with both transforms on the chip
WINOF = 7000001      # F(2x2, 3x3) with the transform domain kept on the chip (csrc/conv_wino_fused.hip): one persistent launch,
                     # no workspace; same transformed filters, same arithmetic (and error family) as a WINO plan


def _tune_tag():
    """Candidate-set tag, part of every tune-cache key: a cache entry written before a family of candidates existed, or in a
    process that had it switched off (SSP_WINOGRAD=0, SSP_WINO_TILES, SSP_WINO_MIN_CHANNELS), must not keep that family
    from ever being timed - and must not hand a Winograd code to a process that switched it off."""
    if os.environ.get('SSP_WINOGRAD', '1') == '0':
        return 'r5-direct'
    return 'r6-w%s-c%s-%s-f%s' % (os.environ.get('SSP_WINO_TILES', '2,4'), os.environ.get('SSP_WINO_MIN_CHANNELS', '128'),
                                  os.environ.get('SSP_WINO4_MIN_CHANNELS', '64'),
                                  os.environ.get('SSP_WINO_FUSED', '1') + os.environ.get('SSP_WGRAD_FUSED_PREFER', '') +
                                  os.environ.get('SSP_ONCHIP_PREFER', '') + os.environ.get('SSP_ONCHIP_CREDIT_MS', ''))


def wino_fused(code):
    """True for the on-chip F(2x2) plan codes (7xxxxxx)."""
    return 7000000 <= code < 8000000


def wino_tile(code):
    """Output-tile edge of a plan code: 2 / 4 for a Winograd plan, 0 for a direct one (ssp_conv_plan_wino_tile).  The on-chip
    F(2x2) codes count as tile 2: same filter transform, same error family."""
    return 2 if (WINO <= code < WINO + 1000000 or wino_fused(code)) else (4 if WINO4 <= code < WINO4 + 1000000 else 0)
BN_EPS = 1e-4        # darknet.py:157
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default


# bumped by writers that change parameter memory without going through torch ops (singleshotpose_amd.optim.SGD's
# fused launch): part of the packed-filter cache key
_WEIGHTS_EPOCH = [0]
# autotuned igemm plan per launch shape, shared by every Plan of the process: multi-scale training (dataset.py:66-90
# draws a new resolution every 10 batches) revisits the same ~20 shapes, each is timed once
_TUNE_CACHE = {}
# ... and per FAMILY of that launch (0 = direct kernels, 2 / 4 = Winograd F(2x2) / F(4x4)): launch shape -> {family: (fastest code
# of the family, its time in ms)} - what the error budget of the forward plans chooses from (Plan._apply_head_budget)
_TUNE_FAMILY = {}
_HEAD_BUDGET_CACHE = {}        # (plan shape, tag, budget, model) -> record: a rebuilt plan re-measures nothing
_HEAD_BUDGET_PINNED = {}       # (plan shape, tag, budget) -> {layer: forward plan code}: decisions read from / written to the
                               # SSP_TUNE_CACHE file - a process started with that file runs the SAME plan set without measuring
                               # (profiled runs hold training steps only; multi-process jobs can share one file)
_TUNE_VERIFIED = {}            # launch shape -> the plan code(s) that passed verify-after-tune in this process
_TUNE_VERIFIED_ALSO = set()    # (launch shape, code) pairs verified besides the chosen one (the other families' fastest codes)
TUNE_REJECTED = []             # (shape key, plan code) pairs verify-after-tune refused
_TUNE_CACHE_FILE = [None]      # SSP_TUNE_CACHE=<json file>: loaded once, rewritten whenever a new shape was timed


def _tune_cache_load():
    path = os.environ.get('SSP_TUNE_CACHE')
    if not path or _TUNE_CACHE_FILE[0] == path:
        return
    _TUNE_CACHE_FILE[0] = path
    if os.path.isfile(path):
        import json
        with open(path) as f:
            for k, v in json.load(f).items():
                if k.startswith('fam|'):
                    _TUNE_FAMILY.setdefault(tuple(json.loads(k[4:])), {int(f_): (int(c), t) for f_, (c, t) in v.items()})
                elif k.startswith('budget|'):      # a pinned error-budget decision (layer -> forward code) of a plan shape
                    _HEAD_BUDGET_PINNED.setdefault(tuple(json.loads(k[7:])), {int(i): int(c) for i, c in v.items()})
                else:
                    _TUNE_CACHE.setdefault(tuple(json.loads(k)), int(v))


def _tune_cache_save():
    path = _TUNE_CACHE_FILE[0]
    if path:
        import json
        tmp = path + '.tmp.%d' % os.getpid()
        with open(tmp, 'w') as f:
            out = {json.dumps(list(k)): v for k, v in _TUNE_CACHE.items()}
            out.update({'fam|' + json.dumps(list(k)): {str(f_): [c, t] for f_, (c, t) in v.items()} for k, v in _TUNE_FAMILY.items()})
            out.update({'budget|' + json.dumps(list(k)): {str(i): c for i, c in v.items()} for k, v in _HEAD_BUDGET_PINNED.items()})
            json.dump(out, f, indent=0, sort_keys=True)
        os.replace(tmp, path)


def _storage_refs(t):
    """Reference count of t's storage (the tensor itself, every live view of it, the Python storage wrapper), or None
    when this torch build does not expose it."""
    fn = getattr(torch._C, '_storage_Use_Count', None)
    return fn(t.untyped_storage()._cdata) if fn is not None else None


def _storage_shared(t, refs_alone, params):
    """True when something besides `t` itself still views t's storage (refs_alone = _storage_refs(t) right after t was
    allocated).  Without the storage counter: true when a parameter's .grad aliases it (views held elsewhere are then
    not seen: the conservative answer would cost a 202 MB allocation per backward)."""
    refs = _storage_refs(t)
    if refs is not None and refs_alone is not None:
        return refs != refs_alone
    sp = t.untyped_storage().data_ptr()
    return any(p.grad is not None and p.grad.untyped_storage().data_ptr() == sp for p in params)


_SIDE_STREAMS = {}


def _side_stream(device):
    """The second stream (filter repacks / transforms in forward, filter gradients in backward), at the HIGH HIP stream
    priority: its launches then win the dispatch slots that free up next to the main stream's, which shortened the step by
    0.2-0.3 ms (0.6 %) in two interleaved A/B rounds (profiles/r03_step_ab_winograd.txt).  SSP_SIDE_PRIORITY=0 restores the
    runtime's default (the device offers 0 and -1).

    ONE per device and process, shared by every plan, and created as early as possible (dist.init_distributed creates it
    BEFORE the RCCL process group): HIP multiplexes its streams over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
    default) in creation order, and a side stream created after RCCL's own streams landed on the main stream's queue -
    the two chains then ran one after the other instead of side by side (single-rank rehearsal, profiles/r04_rccl_queues.txt:
    forward 12.1 ms with no two kernels ever overlapping against 10.1 ms, step 35.7 ms against 28.7 ms)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get('SSP_SIDE_PRIORITY', '-1')))
    return st


def weights_changed():
    _WEIGHTS_EPOCH[0] += 1


def _is_packed(wt, cinp):
    """True when the (Cout,Cin,kh,kw) parameter is stored channels-last, i.e. [Cout][kh][kw][Cin] in memory with no channel
    padding needed: that is the forward operand layout and the layout the filter gradient is accumulated in, so the
    kernels read / write the parameter (and its gradient) in place."""
    return wt.size(1) == cinp and wt.permute(0, 2, 3, 1).is_contiguous()


def _ptr(t, offset=0):
    return t.data_ptr() + 4 * offset


def _pad4(c):
    return (c + 3) // 4 * 4


def choose_under_budget(table, budget):
    """The error-aware choice of Plan._apply_head_budget, as a pure function (tests/test_host.py).

    table: {layer: [(family, plan code, launch ms, head deviation), ...]} - the candidates of every layer, FASTEST FIRST, the
    direct code (deviation 0 by definition: it is the reference the deviations are measured against) among them.  The
    deviations of different layers are independent roundings and add in quadrature.  Starting from the fastest candidate
    everywhere, the layer that buys the most squared deviation per millisecond is moved to its next more accurate candidate
    until sqrt(sum d^2) <= budget (or nothing is left to move).  Returns ({layer: index into its list}, [(layer, family
    left, family taken), ...])."""
    choice = {i: 0 for i in table}
    moved = []
    total = lambda: sum(table[i][choice[i]][3] ** 2 for i in table) ** 0.5
    while total() > budget:
        best, gain = None, 0.0
        for i, rows in table.items():
            k = choice[i]
            for k2 in range(k + 1, len(rows)):
                if rows[k2][3] < rows[k][3]:      # the next candidate that is actually more accurate
                    dt = max((rows[k2][2] or 0.0) - (rows[k][2] or 0.0), 1e-6)
                    g = (rows[k][3] ** 2 - rows[k2][3] ** 2) / dt
                    if g > gain:
                        best, gain = (i, k2), g
                    break
        if best is None:
            break
        moved.append((best[0], table[best[0]][choice[best[0]]][0], table[best[0]][best[1]][0]))
        choice[best[0]] = best[1]
    return choice, moved


class _Act(object):
    """A [pixels][ld] fp32 view: tensor + channel offset."""
    __slots__ = ('t', 'off', 'C', 'H', 'W', 'ld')

    def __init__(self, t, off, C, H, W, ld):
        self.t, self.off, self.C, self.H, self.W, self.ld = t, off, C, H, W, ld

    @property
    def ptr(self):
        return _ptr(self.t, self.off)


class _ConvSpec(object):
    """One conv block of a plan.  `raw` (the block's raw conv output, kept for backward) is allocated on first use: the
    fused first block (csrc/conv_first.hip) recomputes its convolution in every pass and never touches it - 1.42 GB at
    batch 64, 416 x 416 that a plan (and the plan cache's memory budget) does not have to carry."""
    _raw = None
    _raw_spec = None     # (allocator, floats, tensor kwargs)

    @property
    def raw(self):
        if self._raw is None:
            alloc, n, kw = self._raw_spec
            self._raw = alloc(n, **kw)
        return self._raw


class Plan(object):
    """Buffers + launch sequence for one (batch, height, width) input shape."""

    def __init__(self, net, B, H, W, device):
        self.net = net
        self.B, self.H, self.W = B, H, W
        self.device = device
        blocks = net.blocks
        self.shapes = layer_shapes(blocks, W, H)  # (w, h, c) per layer
        nl = len(blocks) - 1
        self.nl = nl

        # ---- consumer analysis (who reads layer l's output) ----
        consumers = [[] for _ in range(nl)]
        for ind, block in enumerate(blocks[1:]):
            t = block['type']
            if t in ('convolutional', 'maxpool', 'reorg'):
                if ind > 0:
                    consumers[ind - 1].append(ind)
            elif t == 'route':
                for l in resolve_layers(block['layers'], ind):
                    consumers[l].append(ind)
            elif t in ('region', 'cost'):
                pass
            else:
                raise NotImplementedError(
                    "block type '%s' is outside the yolo-pose hot path (SURVEY.md section 8); not built" % t)
        self.consumers = consumers
        # the network output is the last non-region layer
        self.last = max(i for i, b in enumerate(blocks[1:]) if b['type'] not in ('region', 'cost'))

        f32 = dict(dtype=torch.float32, device=device)
        in_c = int(blocks[0].get('channels', 3))
        self.in_c = in_c
        self.in_cp = _pad4(in_c)
        self.x_nhwc = torch.empty(B * H * W * self.in_cp, **f32)
        self.input_act = _Act(self.x_nhwc, 0, self.in_cp, H, W, self.in_cp)

        self.convs = {}      # layer index -> _ConvSpec
        self.fused_pool = set()   # maxpool layers folded into the preceding conv block
        self.acts = [None] * nl   # forward outputs
        self.ops_fwd = []
        wsz = dsz = 0
        prev = self.input_act
        for ind, block in enumerate(blocks[1:]):
            t = block['type']
            w, h, c = self.shapes[ind]
            if t == 'convolutional':
                k, s = int(block['size']), int(block['stride'])
                if s != 1 or k not in (1, 3) or (k == 3 and not int(block['pad'])):
                    raise NotImplementedError("conv size=%d stride=%d pad=%s is outside the yolo-pose hot path" % (k, s, block['pad']))
                cs = _ConvSpec()
                cs.ind, cs.k = ind, k
                cs.bn = int(block['batch_normalize']) != 0
                act = block['activation']
                if act not in ('leaky', 'linear', 'relu'):
                    raise NotImplementedError("activation '%s'" % act)
                cs.slope = {'leaky': 0.1, 'linear': 1.0, 'relu': 0.0}[act]
                cs.inp = prev
                cs.first = prev is self.input_act
                cs.cin = in_c if cs.first else prev.C
                cs.cinp = _pad4(cs.cin)
                cs.cout = c
                if prev.ld < cs.cinp:
                    raise NotImplementedError("conv input with a channel count that is not a multiple of 4")
                cs.H, cs.W = prev.H, prev.W
                cs.coutp = _pad4(c)
                seq = net.models[ind]
                cs.conv = seq[0]
                cs.bnm = seq[1] if cs.bn else None
                # fuse a following 2x2/2 max-pool when it is the only consumer
                nxt = blocks[ind + 2] if ind + 2 < len(blocks) else None
                cs.pool = bool(cs.bn and nxt is not None and nxt['type'] == 'maxpool' and int(nxt['size']) == 2 and
                               int(nxt['stride']) == 2 and consumers[ind] == [ind + 1] and cs.H % 2 == 0 and cs.W % 2 == 0)
                M = B * cs.H * cs.W
                cs.M = M
                alloc = torch.empty if cs.coutp == c else torch.zeros
                cs._raw_spec = (alloc, M * cs.coutp, f32)      # raw conv output (kept for backward), see _ConvSpec.raw
                cs.ldraw = cs.coutp
                cs.needs_act = cs.bn or cs.slope != 1.0
                if cs.needs_act:
                    Ho, Wo = (cs.H // 2, cs.W // 2) if cs.pool else (cs.H, cs.W)
                    cs.out = _Act(alloc(B * Ho * Wo * cs.coutp, **f32), 0, c, Ho, Wo, cs.coutp)
                else:
                    cs.out = _Act(cs.raw, 0, c, cs.H, cs.W, cs.coutp)
                cs.plan_fwd = cs.plan_dgrad = 0     # explicit igemm plan codes (0 = library heuristic), see _autotune
                # per-channel vectors: mean, invstd, scale, shift, c1, c2, dgamma, dbeta
                cs.vec = torch.zeros(8, cs.coutp, **f32)
                if not cs.bn:
                    cs.vec[1].fill_(1.0)
                    cs.vec[2].fill_(1.0)
                cs.woff, cs.doff = wsz, dsz
                wsz += c * k * k * cs.cinp
                dsz += cs.cinp * k * k * cs.coutp if not cs.first else 0
                self.convs[ind] = cs
                if cs.pool:
                    self.fused_pool.add(ind + 1)
                    self.acts[ind] = None
                    self.acts[ind + 1] = cs.out
                else:
                    self.acts[ind] = cs.out
                self.ops_fwd.append(('conv', cs))
                prev = cs.out
            elif t == 'maxpool':
                if ind in self.fused_pool:
                    prev = self.acts[ind]
                    continue
                k, s = int(block['size']), int(block['stride'])
                if k != 2 or s != 2:
                    raise NotImplementedError("maxpool size=%d stride=%d (MaxPoolStride1, darknet.py:8-14) is not instantiated by the pose cfgs; not built" % (k, s))
                src = prev
                out = _Act(torch.empty(B * h * w * src.ld, **f32), 0, c, h, w, src.ld)
                self.acts[ind] = out
                self.ops_fwd.append(('maxpool', ind, src, out))
                prev = out
            elif t == 'reorg':
                if int(block['stride']) != 2:
                    raise NotImplementedError("reorg stride != 2")
                src = prev
                out = _Act(torch.empty(B * h * w * c, **f32), 0, c, h, w, c)
                self.acts[ind] = out
                self.ops_fwd.append(('reorg', ind, src, out))
                prev = out
            elif t == 'route':
                layers = resolve_layers(block['layers'], ind)
                srcs = [self.acts[l] for l in layers]
                if any(s is None for s in srcs):
                    raise RuntimeError("route to a layer whose output was fused away")
                if len(layers) == 1:
                    self.acts[ind] = srcs[0]
                    self.ops_fwd.append(('alias', ind, layers[0]))
                else:
                    out = _Act(torch.empty(B * h * w * c, **f32), 0, c, h, w, c)
                    self.acts[ind] = out
                    self.ops_fwd.append(('concat', ind, layers, srcs, out))
                prev = self.acts[ind]
            else:  # region / cost: not executed in forward (darknet.py:119-127)
                self.acts[ind] = prev
        # Filter staging buffers are allocated when first needed: channels-last parameters are used in place, so only the
        # padded first layer (or a filter a caller replaced by a plain contiguous tensor) gets a forward / gradient
        # staging copy, and the data-gradient operands (202 MB) exist only in plans that run a backward.
        self._dpack_floats = max(dsz, 1)
        self._dpack = None
        self._dgrad_tuned = False
        self._dgrad_fallback = False     # Plan.backward's no-tuning fallback made its choices (once per plan)
        self.bn_partial = torch.empty(_lib.query('ssp_bn_bwd_blocks') * 2 * max(cs.coutp for cs in self.convs.values()), **f32)
        self._tune = device.type == 'cuda' and os.environ.get('SSP_AUTOTUNE', '1') != '0'
        if self._tune:
            self._autotune('fwd')
        # statistics / split-K workspaces follow the (tuned or heuristic) plan of each launch
        for cs in self.convs.values():
            M = cs.M
            # First block in training mode: conv + BN + leaky + pool with the convolution recomputed by every pass
            # instead of stored (csrc/conv_first.hip): the 32-channel full-resolution map (1.42 GB at batch 64, the
            # largest tensor of the net, written and re-read five times by the generic path) never exists.
            cs.first_fused = bool(cs.first and cs.bn and cs.pool and cs.k == 3 and cs.cinp == 4 and cs.cout == 32 and
                                  cs.H % 2 == 0 and cs.W % 16 == 0 and M * 16 < (1 << 31) and
                                  (M // 4) * cs.out.ld * 4 < (1 << 31) and device.type == 'cuda' and
                                  os.environ.get('SSP_FIRST_FUSED', '1') != '0')
            cs.first_live = False        # the last forward took the fused path (its backward must as well)
            if cs.first_fused:
                cs.first_groups = _lib.query('ssp_first_groups', B, cs.H, cs.W)
                cs.first_tile = _lib.query('ssp_first_tile_pixels')
                cs.first_partial = torch.empty(cs.first_groups * 64, **f32)
            self._size_layer(cs)
        # split-K partial tiles (13x13 layers): one scratch buffer shared by every conv launch of the plan
        self.ws_floats = max([1] + [cs.ws_fwd for cs in self.convs.values()])
        self.ws = torch.empty(self.ws_floats, **f32)
        self._head_budget_done = False
        self._hb_gains = None
        self._hb_gains_after_forward = False
        self._hb_checked = 0
        self.head_budget = None      # record of Plan._apply_head_budget (what was measured, what was chosen)
        self.bn_momentum = BN_MOMENTUM
        self.wversion = {}
        self.wino_version = {}
        self.bnversion = {}
        self.generation = 0
        # flat gradient buffer layout, in backward (reverse layer) order so that all-reduce buckets close early:
        # per conv block: weight | bias  or  weight | bn.weight | bn.bias   (each 16-byte aligned)
        self.grad_layout = {}   # id(param) -> (offset, numel, shape)
        goff = 0
        for ind in sorted(self.convs.keys(), reverse=True):
            cs = self.convs[ind]
            plist = [cs.conv.weight]
            if cs.conv.bias is not None:
                plist.append(cs.conv.bias)
            if cs.bn:
                plist += [cs.bnm.weight, cs.bnm.bias]
            cs.grad_lo = goff
            for prm in plist:
                self.grad_layout[id(prm)] = (goff, prm.numel(), tuple(prm.shape))
                goff += (prm.numel() + 3) // 4 * 4
            cs.grad_hi = goff
        self.grad_total = goff
        self.reducer = None      # singleshotpose_amd.dist.GradReducer (multi-GPU): notified as layers finish
        self.side_stream = None
        self.serial_backward = False   # measurement aid (bench.py): filter gradients on the MAIN stream, so that every
                                       # backward launch runs alone and its HIP-event duration is kernel-exclusive
        self.dgrad_ready = None
        self.grads = {}      # layer index -> _Act gradient buffers, allocated on first backward
        self.out_act = self.acts[self.last]
        self.consumed = False
        self._graph = self._graph_key = self._x_static = self._y_static = None
        self._graph_failed = False

    def _size_layer(self, cs):
        """Statistics / split-K bookkeeping of one conv block for its CURRENT forward plan code (after tuning, and again
        when the error budget moves the layer to another family): tile height of the per-tile BatchNorm partials (0 = the
        counted format of a Winograd launch), their count, the launch's workspace need; the statistics buffer is sized
        for the largest format any timed family of the layer uses, so a plan change never re-allocates it."""
        B = self.B
        q = lambda name, code: _lib.query(name, B, cs.H, cs.W, cs.cinp, cs.cout, cs.k, code)
        cs.tile_m = q('ssp_conv_stats_tile_m', cs.plan_fwd)
        cs.ws_fwd = q('ssp_conv_workspace_floats', cs.plan_fwd)
        cs.ntile = q('ssp_conv_stats_tiles', cs.plan_fwd)
        codes = set([cs.plan_fwd, 0] + [c for c, _ in (getattr(cs, 'fwd_fams', None) or {}).values()])
        nstat = max(q('ssp_conv_stats_floats', c) for c in codes)
        if getattr(cs, 'first_fused', False):
            nstat = max(nstat, cs.first_groups * 64)
        if cs.bn and (getattr(cs, 'stats', None) is None or cs.stats.numel() < nstat):
            cs.stats = torch.empty(nstat, dtype=torch.float32, device=self.device)

    def _apply_head_budget(self):
        """Error-aware admission of the forward plans: a NETWORK-level rounding budget, measured on the live batch.

        A Winograd F(4x4) launch carries ~2.5e-7 (rms, of the output's range) of rounding that no summation order removes
        - every M = sum_c U.V element is ONE fp32 number 3-4 x the size of the output it is cancelled into
        (tools/wino_error_budget.py) - against 4e-8 for the direct kernel with chunked accumulation and ~1e-7 for F(2x2).
        What that costs at the network output depends on WHERE the layer sits: BatchNorm re-normalises every block, so a
        perturbation of an early block's output is amplified layer after layer (measured in float64 on the CPU,
        tools/head_amplification.py: white noise of 1e-6 of the range on layer 4's raw output moves the head by 1.7e-4 of its
        range, on layer 23's by 6e-6) - F(4x4) on the 104 x 104 and 52 x 52 layers IS the head error of the step (6e-5 of
        the 1e-4 bar), F(4x4) on the 13 x 13 layers, where it saves the most time, costs nothing measurable.

        So, once per plan, on the first training batch: one forward with every eligible layer on its fastest DIRECT code
        gives the reference head; then, layer by layer, one forward with that layer alone on its Winograd candidate
        gives d_l = max|head - reference| / max|reference| - the head deviation that candidate is responsible for, on the
        live operands and weights.  The deviations of different layers are independent roundings: they add in quadrature.
        Starting from the fastest code everywhere, the layer with the largest (d^2 saved per millisecond lost) is moved to
        its next more accurate family (F(4x4) -> F(2x2) -> direct) until sqrt(sum d_l^2) <= SSP_HEAD_ERR_BUDGET (default
        3.5e-5 of the head's range; 0 = off: always the fastest code).  BatchNorm running statistics are not touched by the
        measurement forwards (momentum 0).  The record is kept in `plan.head_budget` (bench.py prints it)"""
        self._head_budget_done = True
        self._hb_gains = None
        self._hb_gains_after_forward = False
        self._hb_checked = self.generation
        budget = float(os.environ.get('SSP_HEAD_ERR_BUDGET', '3.5e-5'))
        if budget <= 0 or not self._tune:
            return
        cand = [cs for cs in self.convs.values() if wino_tile(cs.plan_fwd) and 0 in (getattr(cs, 'fwd_fams', None) or {})]
        if not cand:
            return
        # the NETWORK is part of the key: the deviations are properties of this stack of layers (another cfg at the same input
        # shape - tiny-pose against yolo-pose, the multi-object cfg - must not run under this one's budget decisions)
        import zlib
        net_fp = zlib.crc32(repr([(i, cs.cin, cs.cout, cs.k, cs.H, cs.W, bool(cs.bn), bool(cs.pool))
                                  for i, cs in sorted(self.convs.items())]).encode())
        ckey = (self.B, self.H, self.W, _tune_tag(), budget, id(self.net))
        pkey = (self.B, self.H, self.W, _tune_tag(), budget, net_fp)
        rec = _HEAD_BUDGET_CACHE.get(ckey)
        pinned = _HEAD_BUDGET_PINNED.get(pkey)
        if rec is None and pinned is not None and _TUNE_CACHE_FILE[0] and all(
                any(c == pinned.get(cs.ind) for c, _ in cs.fwd_fams.values()) for cs in cand):
            # (entry -1 of a pinned record: the measured deviation of the chosen plans, in units of 1e-12)
            rec = dict(budget=budget, head_deviation=pinned.get(-1, 0) * 1e-12 if -1 in pinned else float('nan'),
                       fastest={cs.ind: cs.plan_fwd for cs in cand}, chosen={i: c for i, c in pinned.items() if i >= 0},
                       cost_ms=float('nan'), moved=[(cs.ind, wino_tile(cs.plan_fwd), wino_tile(pinned[cs.ind])) for cs in cand
                                                    if pinned[cs.ind] != cs.plan_fwd], table={}, pinned=True)
            _HEAD_BUDGET_CACHE[ckey] = rec
        if rec is None:
            o = self.out_act
            head = lambda: o.t[o.off:o.off + self.B * o.H * o.W * o.ld].clone()
            fastest = {cs.ind: cs.plan_fwd for cs in cand}

            def run(assign):
                for cs in cand:
                    cs.plan_fwd = assign[cs.ind]
                    self._size_layer(cs)
                self._fit_workspace()
                self._forward_body(True, False, False)
                return head()
            mom, self.bn_momentum = self.bn_momentum, 0.0
            try:
                direct = {cs.ind: cs.fwd_fams[0][0] for cs in cand}
                ref = run(direct)
                den = max(float(ref.abs().max()), 1e-30)
                table = {}         # layer -> [(family, code, ms, deviation)], fastest first; the direct entry has deviation 0
                for cs in cand:
                    rows = []
                    for f_, (c_, t_) in cs.fwd_fams.items():
                        if f_ == 0:
                            rows.append((0, c_, t_, 0.0))
                            continue
                        if t_ is not None and cs.fwd_fams[0][1] is not None and t_ >= cs.fwd_fams[0][1]:
                            continue      # slower than the direct code: never an option
                        assign = dict(direct)
                        assign[cs.ind] = c_
                        d = float((run(assign) - ref).abs().max()) / den
                        rows.append((f_, c_, t_, d))
                    # fastest first; the code the tuner actually chose (near-tie rule) leads among equals, so that `moved`
                    # and `cost_ms` describe moves the layer really makes
                    rows.sort(key=lambda r: (0 if r[1] == fastest[cs.ind] else 1, r[2] if r[2] is not None else 0.0))
                    table[cs.ind] = rows
            finally:
                self.bn_momentum = mom
            choice, moved = choose_under_budget(table, budget)
            total = lambda: sum(table[i][choice[i]][3] ** 2 for i in table) ** 0.5
            rec = dict(budget=budget, head_deviation=total(), fastest=fastest,
                       chosen={i: table[i][choice[i]][1] for i in table},
                       cost_ms=sum((table[i][choice[i]][2] or 0.0) - (table[i][0][2] or 0.0) for i in table),
                       moved=moved, table={i: [(f_, c_, None if t_ is None else round(t_, 4), float('%.3g' % d))
                                               for f_, c_, t_, d in rows] for i, rows in table.items()})
            _HEAD_BUDGET_CACHE[ckey] = rec
            if _TUNE_CACHE_FILE[0]:
                _HEAD_BUDGET_PINNED[pkey] = dict(rec['chosen'])
                _HEAD_BUDGET_PINNED[pkey][-1] = int(round(rec['head_deviation'] * 1e12))
                _tune_cache_save()
        for cs in cand:
            cs.plan_fwd = rec['chosen'].get(cs.ind, cs.plan_fwd)
            self._size_layer(cs)
            cs.wino_u = {t: b for t, b in (getattr(cs, 'wino_u', None) or {}).items() if t == wino_tile(cs.plan_fwd)}
        self._fit_workspace()
        self.head_budget = rec
        # the amplification the decisions were measured under (head_budget_drifted).  A record adopted from the cache ran no
        # measurement forward here: the scale vectors of this batch exist only after the forward body (forward() takes them)
        self._hb_gains = None if rec.get('pinned') else self._bn_gains()
        self._hb_gains_after_forward = bool(rec.get('pinned'))

    # The deviations the budget admits a plan under are measured ONCE, on the weights and BatchNorm statistics of the plan's
    # first training batch - and the amplification of a layer's rounding on the way to the head is a product of BatchNorm gains
    # (|gamma| / sigma per block), which training moves.  Two triggers re-measure:
    #   * load_weights / load_weights_until_last (Darknet._load_blocks): another network as far as rounding goes;
    #   * every SSP_HEAD_BUDGET_EVERY (default 256) training forwards the largest |gamma * invstd| of every BatchNorm block is
    #     compared with its value at measurement time (one stacked reduction, one 88-byte read): a block that moved by more
    #     than 2x either way invalidates the record.  The re-measurement costs ~0.3 s (<= 27 forwards), like the first one.
    def head_budget_stale(self):
        if self._head_budget_done:
            self._head_budget_done = False
            for k in [k for k in _HEAD_BUDGET_CACHE if k[:3] == (self.B, self.H, self.W) and k[-1] == id(self.net)]:
                _HEAD_BUDGET_CACHE.pop(k, None)
            # (a pinned decision read from SSP_TUNE_CACHE belongs to the weights it was measured on, too)
            for k in [k for k in _HEAD_BUDGET_PINNED if k[:3] == (self.B, self.H, self.W)]:
                _HEAD_BUDGET_PINNED.pop(k, None)
            # every timed family is a candidate again
            for cs in self.convs.values():
                fams = getattr(cs, 'fwd_fams', None)
                if fams:
                    best = min((v for v in fams.values() if v[1] is not None), key=lambda v: v[1], default=None)
                    if best is not None and best[0] != cs.plan_fwd:
                        cs.plan_fwd = best[0]
                        self._size_layer(cs)
            self._fit_workspace()

    def _bn_gains(self):
        bn = [cs for _, cs in sorted(self.convs.items()) if cs.bn]
        if not bn:
            return None
        return torch.stack([cs.vec[2][:cs.cout].abs().max() for cs in bn]).cpu()

    def head_budget_drifted(self):
        """True when a BatchNorm block's largest |gamma * invstd| moved by more than 2x since the budget was measured."""
        if self._hb_gains is None:
            return False
        now = self._bn_gains()
        r = (now / self._hb_gains.clamp_min(1e-30)).clamp_min(1e-30)
        return bool(((r > 2.0) | (r < 0.5)).any())

    def _sync_codes(self, stage):
        """Multi-GPU: adopt rank 0's plan codes (singleshotpose_amd.dist.sync_plans installs the hook on the model).
        stage 0: forward codes (called once, on the plan's first training forward, after the head-error budget);
        stage 1: data-gradient codes and the filter gradients' direct / Winograd choice (after their tuning)."""
        fn = getattr(self.net, '_plan_sync', None)
        if fn is None:
            return
        order = sorted(self.convs)
        if stage == 0:
            mine = [self.convs[i].plan_fwd for i in order]
        else:
            mine = [self.convs[i].plan_dgrad for i in order] + [getattr(self.convs[i], 'wgrad_wino', 0) for i in order]
        head = [self.B, self.H, self.W, stage, len(order)]
        got = fn(head + mine)
        # Adopt only when EVERY rank can: same plan shape everywhere, and every code one this rank's environment allows
        # (SSP_WINOGRAD / SSP_WINO_TILES / SSP_WINO_FUSED may differ between ranks' launch scripts).  Otherwise all ranks
        # fall back to the library's heuristic plans (code 0) for this stage - deterministic and the same everywhere.
        ok = got[:5] == head and len(got) == len(head) + len(mine)
        if ok:
            wino_on = os.environ.get('SSP_WINOGRAD', '1') != '0'
            tiles = tuple(int(t) for t in os.environ.get('SSP_WINO_TILES', '2,4').split(',') if t)
            fused_on = os.environ.get('SSP_WINO_FUSED', '1') != '0'
            for k, v in enumerate(got[5:]):
                if v == mine[k] or v == 0:
                    continue
                wg = stage == 1 and k >= len(order)                 # (the filter gradients' entries ARE tile sizes)
                t = (2 if v == WGRAD_FUSED else v) if wg else wino_tile(v)
                if t and (not wino_on or t not in tiles or ((v == WGRAD_FUSED if wg else wino_fused(v)) and not fused_on)):
                    ok = False
        all_ok = getattr(fn, 'all_ok', None)
        if all_ok is not None:
            ok = all_ok(ok)
        vals = got[5:] if ok else [0] * len(mine)
        for k, i in enumerate(order):
            cs = self.convs[i]
            if stage == 0:
                if vals[k] != cs.plan_fwd:
                    cs.plan_fwd = vals[k]
                    self._size_layer(cs)
            else:
                cs.plan_dgrad = vals[k]
                w = vals[len(order) + k]
                if w != getattr(cs, 'wgrad_wino', 0):
                    cs.wgrad_wino = w
                    if w:
                        cs.wino_ws_floats = _lib.query('ssp_conv_wgrad_wino_workspace_floats_t', self.B, cs.H, cs.W, cs.cinp, cs.cout, w)
                    cs.wino_ws = None
        if stage == 0:
            self._fit_workspace()

    def _fit_workspace(self):
        need = max([1] + [cs.ws_fwd for cs in self.convs.values()] + [getattr(cs, 'ws_dgrad', 0) for cs in self.convs.values()])
        if need > self.ws_floats:
            torch.cuda.current_stream().synchronize()      # nothing in flight may still use the old workspace
            self.ws_floats = need
            self.ws = torch.empty(need, dtype=torch.float32, device=self.device)
            self._graph = None          # a captured inference chain holds the old workspace pointer

    def footprint(self):
        """Bytes of device memory this plan's forward buffers hold or will hold after a training-mode forward (each storage
        counted once; raw conv outputs are allocated on first use - every block's but the fused first one's count here)."""
        seen, total = set(), 0
        ts = [self.x_nhwc, self.ws, self.bn_partial] + [a.t for a in self.acts if a is not None]
        for cs in self.convs.values():
            ts += [cs._raw, cs.vec, getattr(cs, 'stats', None), getattr(cs, 'first_partial', None)]
            if cs._raw is None and not getattr(cs, 'first_fused', False):
                total += cs._raw_spec[1] * 4
        for t in ts:
            if t is not None and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                total += t.numel() * t.element_size()
        return total

    def nbytes_now(self):
        """Bytes of device memory the plan holds right now: forward buffers, and whatever the first backward and the tuner
        added since (gradient buffers, data-gradient operands, Winograd filters and per-layer workspaces)."""
        seen, total = set(), 0
        ts = [self.x_nhwc, self.ws, self.bn_partial, self._dpack] + [a.t for a in self.acts if a is not None]
        ts += [g.t for g in self.grads.values()]
        for cs in self.convs.values():
            ts += [cs._raw, cs.vec, getattr(cs, 'stats', None), getattr(cs, 'first_partial', None), getattr(cs, 'wbuf', None),
                   getattr(cs, 'gbuf', None), getattr(cs, 'wino_ws', None), getattr(cs, 'first_wpart', None)]
            ts += list((getattr(cs, 'wino_u', None) or {}).values()) + list((getattr(cs, 'wino_ud', None) or {}).values())
            bnp = getattr(cs, 'bnp', None)
            if bnp is not None:
                ts.append(bnp[0])
        for t in ts:
            if t is not None and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                total += t.numel() * t.element_size()
        return total

    # ------------------------------------------------------------------ lazily allocated filter staging
    def _wbuf(self, cs):
        if getattr(cs, 'wbuf', None) is None:
            cs.wbuf = torch.empty(cs.cout * cs.k * cs.k * cs.cinp, dtype=torch.float32, device=self.device)
        return cs.wbuf

    def _wino_u(self, cs, tile):
        """Winograd-transformed forward filters [(tile+2)^2][Cout][Cin] of a layer whose forward plan is a Winograd code of
        that tile size (one buffer per tile size; the tuner drops the ones it did not choose)."""
        if getattr(cs, 'wino_u', None) is None:
            cs.wino_u = {}
        if tile not in cs.wino_u:
            cs.wino_u[tile] = torch.empty((tile + 2) ** 2 * cs.cout * cs.cinp, dtype=torch.float32, device=self.device)
        return cs.wino_u[tile]

    def _wino_ud(self, cs, tile):
        """... and of the data-gradient operand [(tile+2)^2][Cin][Cout] (from the flipped / transposed [Cin][tap][Cout] layout)."""
        if getattr(cs, 'wino_ud', None) is None:
            cs.wino_ud = {}
        if tile not in cs.wino_ud:
            cs.wino_ud[tile] = torch.empty((tile + 2) ** 2 * cs.cin * cs.coutp, dtype=torch.float32, device=self.device)
        return cs.wino_ud[tile]

    def _wino_ws(self, cs):
        """Per-layer Winograd workspace of a layer whose FILTER GRADIENT runs in the Winograd domain (V | dM | dU, on the
        side stream).  When the layer's forward plan is a Winograd code too, the training forward is handed this same
        buffer (V | M): the transformed input V it leaves at the head is what the filter gradient needs, so the layer
        input is transformed once per step (csrc/conv_wgrad.hip ssp_conv_wgrad_wino_launch, x == NULL)."""
        if getattr(cs, 'wino_ws', None) is None:
            n = cs.wino_ws_floats
            if wino_tile(cs.plan_fwd) and not wino_fused(cs.plan_fwd):
                n = max(n, _lib.query('ssp_conv_workspace_floats', self.B, cs.H, cs.W, cs.cinp, cs.cout, cs.k, cs.plan_fwd))
            cs.wino_ws = torch.empty(n, dtype=torch.float32, device=self.device)
        return cs.wino_ws

    def _gbuf(self, cs):
        if getattr(cs, 'gbuf', None) is None:
            cs.gbuf = torch.empty(cs.cout * cs.k * cs.k * cs.cinp, dtype=torch.float32, device=self.device)
        return cs.gbuf

    def _prepare_backward(self, tune=True):
        """First forward that will be followed by a backward: data-gradient operand buffer, dgrad plan tuning, and the
        split-K workspace those plans need.

        tune=False (Plan.backward's fallback, entered when the forward ran without gradient bookkeeping): the saved raw
        conv outputs are live - timing / verify-after-tune launches would overwrite them (they randomise their operands)
        - so the data-gradient launches only take choices that were timed AND verified earlier in this process, else the
        library's heuristic; a later forward with need_grad tunes them.  The fallback's own choices are made once per plan
        (not per backward)."""
        if self._dpack is None:
            self._dpack = torch.empty(self._dpack_floats, dtype=torch.float32, device=self.device)
        if self._dgrad_tuned or (not tune and self._dgrad_fallback):
            return
        self._dgrad_tuned = tune
        self._dgrad_fallback = True
        if self._tune and tune:
            self._autotune('dgrad')
            self._tune_wgrad()
        elif self._tune:
            wino_on = os.environ.get('SSP_WINOGRAD', '1') != '0'
            for cs in self.convs.values():
                key = self._dgrad_key(cs)
                # (the code that was verified, not just the key: a cached Winograd code that SSP_WINOGRAD=0 kept out of
                # this process must not come back through this fallback)
                code = _TUNE_CACHE.get(key, 0)
                cs.plan_dgrad = code if _TUNE_VERIFIED.get(key) == code and (not wino_tile(code) or wino_on) else 0
                cs.wgrad_wino = 0
        if tune:
            self._sync_codes(1)              # multi-GPU: ... and rank 0's data- / filter-gradient choices
        need = 1
        for cs in self.convs.values():
            cs.ws_dgrad = 0 if cs.first else _lib.query('ssp_conv_workspace_floats', self.B, cs.H, cs.W, cs.coutp,
                                                        cs.cin, cs.k, cs.plan_dgrad)
            need = max(need, cs.ws_dgrad)
        self._plan_bn_fusion()
        if need > self.ws_floats:
            torch.cuda.current_stream().synchronize()      # nothing in flight may still use the old workspace
            self.ws_floats = need
            self.ws = torch.empty(need, dtype=torch.float32, device=self.device)
            self._graph = None          # a captured inference chain holds the old workspace pointer

    def _wgrad_key(self, cs):
        return ('wgrad', self.B, cs.H, cs.W, cs.cinp, cs.cout, cs.ldraw, cs.inp.ld, _tune_tag())

    def _tune_wgrad(self):
        """Filter gradients of the deep 3x3 layers: direct kernel or Winograd domain (ssp_conv_wgrad_wino_t, tile 2 or 4),
        whichever is fastest on this shape; a Winograd form is admitted only after its result on seeded operands agrees with
        the direct kernel's to 3e-5 of the gradient's range.  Runs before the first forward of the plan (its operands are the
        plan's own, still empty, buffers).  The choice (0 = direct, else the tile size) is cached per launch shape like the
        igemm plans."""
        call = _lib.call
        st = torch.cuda.current_stream().cuda_stream
        wino_on = os.environ.get('SSP_WINOGRAD', '1') != '0'
        verify = os.environ.get('SSP_TUNE_VERIFY', '1') != '0'
        tiles = tuple(int(t) for t in os.environ.get('SSP_WINO_TILES', '2,4').split(',') if t)
        gen = torch.Generator(device=self.device)
        gen.manual_seed(4321)
        n_known = len(_TUNE_CACHE)
        for cs in self.convs.values():
            cs.wgrad_wino = 0
            cmin = min(cs.cin, cs.cout)
            if not (wino_on and cs.k == 3 and not cs.first and cs.cinp == cs.cin and cs.coutp == cs.cout and
                    cs.cin % 16 == 0 and cs.cout % 16 == 0):
                continue
            # tile sizes worth timing: F(2x2) from SSP_WINO_MIN_CHANNELS (128) channels on both sides, F(4x4) from 64 ...
            ts_ok = [t for t in tiles if cmin >= 64 and cmin >= int(os.environ.get('SSP_WINO_MIN_CHANNELS', '128') if t == 2 else
                                                                    os.environ.get('SSP_WINO4_MIN_CHANNELS', '64')) and
                     self.B * ((cs.H + t - 1) // t) * ((cs.W + t - 1) // t) >= 16]
            # ... and F(2x2) with both transforms on the chip (WGRAD_FUSED: csrc/conv_wino_wgrad_fused.hip), 32-channel granularity
            if (2 in tiles and os.environ.get('SSP_WINO_FUSED', '1') != '0' and cs.cin % 32 == 0 and cs.cout % 32 == 0 and
                    (self.B * cs.H * cs.W + cs.W + 1) * max(cs.inp.ld, cs.ldraw) * 4 < (1 << 31)):
                ts_ok.append(WGRAD_FUSED)
            if not ts_ok:
                continue
            key = self._wgrad_key(cs)
            wsn = {t: _lib.query('ssp_conv_wgrad_wino_workspace_floats_t', self.B, cs.H, cs.W, cs.cinp, cs.cout, t) for t in ts_ok}
            cached = _TUNE_CACHE.get(key)
            if cached is not None and (cached == 0 or (cached in ts_ok and (not verify or _TUNE_VERIFIED.get(key) == cached))):
                cs.wgrad_wino = cached
            else:
                ws = torch.empty(max(wsn.values()), dtype=torch.float32, device=self.device)
                dw = [torch.zeros(cs.cout * 9 * cs.cinp, dtype=torch.float32, device=self.device) for _ in range(2)]
                a = cs.inp
                a.t.view(-1, a.ld)[:, a.off % a.ld:a.off % a.ld + cs.cinp].uniform_(-1.0, 1.0, generator=gen)
                cs.raw.view(-1, cs.ldraw)[:, :cs.cout].uniform_(-1.0, 1.0, generator=gen)

                def direct(out, cs=cs):
                    call('ssp_conv_wgrad', cs.raw.data_ptr(), cs.inp.ptr, out.data_ptr(), self.B, cs.H, cs.W, cs.cinp, cs.cout,
                         cs.ldraw, cs.inp.ld, cs.k, st)

                def wino(out, t, cs=cs, ws=ws):
                    call('ssp_conv_wgrad_wino_t', cs.raw.data_ptr(), cs.inp.ptr, out.data_ptr(), self.B, cs.H, cs.W, cs.cinp,
                         cs.cout, cs.ldraw, cs.inp.ld, t, ws.data_ptr(), ws.numel(), st)

                def timed(fn):
                    fn(torch.empty_like(dw[0]))
                    best = None
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        fn(torch.empty_like(dw[0]))
                        e1.record()
                        e1.synchronize()
                        t = e0.elapsed_time(e1)
                        best = t if best is None else min(best, t)
                    return best

                use, t_best = 0, None
                try:
                    t_best = timed(direct)
                    dw[0].zero_()
                    direct(dw[0])
                    den = float(dw[0].abs().max())
                except _lib.SspError:
                    t_best, den = None, 1.0
                for t in ts_ok:
                    if t_best is None:
                        break
                    try:
                        tt = timed(lambda out, t=t: wino(out, t))
                        if t == WGRAD_FUSED:
                            # timed alone, the HBM-bound transform passes of the other Winograd forms run at full bandwidth; in the
                            # step they share it with the data-gradient stream.  SSP_WGRAD_FUSED_PREFER (default 0.8) credits the on-chip
                            # form (no such passes) with that difference: same-box A/B of the whole step, two interleaved rounds, 26.64 / 26.45 ms
                            # with 1.0 (layers 4 / 6 stay on the F(4x4) chain) against 26.12 / 26.13 ms with 0.8 (profiles/r06_step_ab.txt)
                            # Big launches only (>= SSP_ONCHIP_CREDIT_MS, 0.25 ms alone): at batch 8 the on-chip launches are 0.15 ms
                            # of mostly fixed cost (1024 waves x 144 atomics of flush) and the credit picked them for eight layers -
                            # 6.49 / 6.48 ms per step against 6.33 / 6.32 without it (same box)
                            if tt >= float(os.environ.get('SSP_ONCHIP_CREDIT_MS', '0.25')):
                                tt *= float(os.environ.get('SSP_WGRAD_FUSED_PREFER', '0.8'))
                        if not tt < 0.985 * t_best:
                            continue
                        # verification pair: one accumulation each into zeroed buffers
                        dw[1].zero_()
                        wino(dw[1], t)
                        err = float((dw[0] - dw[1]).abs().max())
                        ok = err <= 3e-5 * max(den, 1e-30)
                    except _lib.SspError:
                        continue
                    if not ok:
                        TUNE_REJECTED.append((key, 'wgrad_wino%d' % t))
                        import warnings
                        warnings.warn("singleshotpose_amd: verify-after-tune refused the Winograd F(%dx%d) filter gradient "
                                      "for %s" % (t, t, key))
                        continue
                    use, t_best = t, tt
                _TUNE_CACHE[key] = use
                _TUNE_VERIFIED[key] = use
                cs.wgrad_wino = use
                del ws, dw
            if cs.wgrad_wino:
                cs.wino_ws_floats = wsn[cs.wgrad_wino]
        torch.cuda.synchronize()
        if len(_TUNE_CACHE) != n_known:
            _tune_cache_save()
        # the timing / verification scratch (workspaces for the deepest candidate, gradient scratch, twin outputs) goes back
        # to the driver: left in the caching allocator it sits reserved next to the plan's own buffers (multi-scale soak)
        torch.cuda.empty_cache()

    def _dgrad_key(self, cs):
        return ('dgrad', self.B, cs.H, cs.W, cs.coutp, cs.cin, cs.k, cs.ldraw, cs.inp.ld, _tune_tag())

    def _plan_bn_fusion(self):
        """BatchNorm-backward reductions folded into the producing data-gradient launch (ssp_conv_dgrad_bnbwd).

        Block `src` qualifies when its (un-pooled) BN + leaky output feeds exactly one consumer and that consumer is a
        conv: the consumer's dgrad writes g = dL/d out(src) - the tile it just finished holds g, one extra read of
        src's raw conv output at the same positions gives sum(dy) and sum(dy * xhat) per tile, and the separate reduce
        pass over raw + g (bn_act_bwd_reduce_kernel: a third of the step's BatchNorm-backward traffic, on the critical
        stream) disappears.  Pooled blocks, blocks with two consumers (route sources) and blocks consumed through a
        route / reorg keep the two-pass form.  SSP_BN_FUSE=0 turns it off (A/B)."""
        on = os.environ.get('SSP_BN_FUSE', '1') != '0'
        for cs in self.convs.values():
            cs.bn_fuse_src = None
            cs.bnp = None
        if not on:
            return
        for cs in self.convs.values():
            if cs.first:
                continue
            src = None
            for i, a in enumerate(self.acts):
                if a is cs.inp:
                    src = i
                    break
            scs = self.convs.get(src)
            if (scs is None or scs.pool or not scs.bn or not scs.needs_act or scs.out is not cs.inp or
                    self.consumers[src] != [cs.ind] or scs.coutp != scs.cout or cs.cin != scs.cout or
                    cs.inp.ld != scs.ldraw or cs.inp.off != 0 or cs.cin % 4):
                continue
            ntile = _lib.query('ssp_conv_stats_tiles', self.B, cs.H, cs.W, cs.coutp, cs.cin, cs.k, cs.plan_dgrad)
            rows = min(ntile, 1024)       # more tiles than rows: folded with atomics into a buffer that stays zeroed
            cs.bn_fuse_src = scs
            scs.bnp = (torch.zeros(rows * scs.cout * 2, dtype=torch.float32, device=self.device), rows, ntile > rows)

    def _repack_dgrad(self, cs, stream):
        src = cs.conv.weight.detach()
        if not cs.packed:
            with torch.cuda.stream(stream):
                src = src.contiguous()       # the plain-layout kernel reads (Cout,Cin,kh,kw) order
                src.record_stream(stream)
        _lib.call('ssp_repack_dgrad_packed' if cs.packed else 'ssp_repack_dgrad', src.data_ptr(),
                  _ptr(self._dpack, cs.doff), cs.cout, cs.cin, cs.coutp, cs.k, stream.cuda_stream)
        tile = wino_tile(cs.plan_dgrad)
        if tile:      # Winograd data-gradient plan: its filter operand is the transform of that layout
            _lib.call('ssp_wino_filter_transform_t', _ptr(self._dpack, cs.doff), self._wino_ud(cs, tile).data_ptr(), cs.cin,
                      cs.coutp, tile, stream.cuda_stream)

    # ------------------------------------------------------------------ per-shape tile / split selection
    def _autotune(self, which):
        """Times the candidate igemm plans (tile rows x split-K x LDS ring depth) of every eligible conv launch of this
        input shape on the real buffers and keeps the fastest (per-shape tile selection).
        The library's shape heuristic is within ~5-15 % of the best choice on some layers; which plan wins depends on
        how the grid fills the 256 CUs.  One-off cost per (B,H,W): ~1 s for yolo-pose.cfg.  SSP_AUTOTUNE=0 disables;
        SSP_TUNE_CACHE=<file> keeps the timed choices across processes (JSON).

        Verify-after-tune: a non-default plan is admitted only after its output on seeded random operands agrees with the
        default plan's output of the same launch (max|a-b| / max|b| <= 1e-5: the plans differ in fp32 summation order
        only) and, for BN layers, after the batch statistics finalized from its per-tile partials agree too.  A code that
        fails is refused (plan 0 runs) and reported in `engine.TUNE_REJECTED`.  Cached choices (this process or the JSON
        file) are re-verified once per process."""
        B = self.B
        call = _lib.call
        st = torch.cuda.current_stream().cuda_stream
        f32 = dict(dtype=torch.float32, device=self.device)
        _tune_cache_load()
        n_known = len(_TUNE_CACHE)
        # tile rows x split-K x ring depth; 3xxxxx / 2xxxxx = hybrid launches (whole resident waves un-split, the tiles of
        # the last partial wave split 3 / 2 ways over K)
        cands = (12813, 12814, 6414, 6413, 12824, 12834, 306413, 306414, 312813, 312814, 206413, 212814,
                 12823, 6423, 6424, 206414, 212813, 406413, 406414, 412813)
        # ring depth 8 = the latency form of the kernel (one workgroup per CU, 7 chunks in flight): only worth timing on
        # grids that cannot give a CU several workgroups anyway (small-batch inference)
        lat = (6418, 12818, 6428, 12828, 6438)
        elig = [cs for cs in self.convs.values() if cs.cinp % 16 == 0]
        if not elig:
            return
        # Winograd F(2x2, 3x3) candidates (csrc/conv_wino.hip): 16/36 of the multiplies at the price of two HBM-bound
        # transform passes - timed against the direct plans on the 3x3 layers with >= 128 channels on both sides (with 64
        # the tuner never picked one: the transforms of the wide maps cost more than the short-K GEMMs save; measured,
        # profiles/r03_step_ab_winograd.txt).  SSP_WINOGRAD=0 turns them off, SSP_WINO_MIN_CHANNELS moves the threshold.
        # F(4x4, 3x3) candidates (WINO4): 36/144 of the multiplies and 2.25 instead of 4 transform floats per pixel, so they
        # are timed from 64 channels up (SSP_WINO_TILES=2 or =4 restricts the tile sizes tried).
        gemm_cands = (6413, 6414, 12813, 12814)
        wino_on = os.environ.get('SSP_WINOGRAD', '1') != '0'
        wino_min = int(os.environ.get('SSP_WINO_MIN_CHANNELS', '128'))
        wino_tiles = tuple(int(t) for t in os.environ.get('SSP_WINO_TILES', '2,4').split(',') if t)

        def wino_bases(cs, which):
            """Plan-code bases (WINO / WINO4) of the Winograd forms worth timing on this launch."""
            if not (wino_on and cs.k == 3 and not cs.first and cs.cinp == cs.cin and cs.coutp == cs.cout and
                    cs.cin % 16 == 0 and cs.cout % 16 == 0):
                return ()
            bases = []
            if 2 in wino_tiles and min(cs.cin, cs.cout) >= wino_min:
                bases.append(WINO)
            if 4 in wino_tiles and min(cs.cin, cs.cout) >= int(os.environ.get('SSP_WINO4_MIN_CHANNELS', '64')):
                bases.append(WINO4)
            return tuple(bases)

        fused_on = wino_on and os.environ.get('SSP_WINO_FUSED', '1') != '0' and 2 in wino_tiles

        def fused_ok(cs, which):
            """The on-chip F(2x2) kernel (csrc/conv_wino_fused.hip) fits: 3x3, channel counts multiples of 32 on both sides of
            the launch (forward: Cin -> Cout; data gradient: Cout -> Cin), no channel padding."""
            return (fused_on and cs.k == 3 and not cs.first and cs.cinp == cs.cin and cs.coutp == cs.cout and
                    cs.cin % 32 == 0 and cs.cout % 32 == 0 and B * cs.H * cs.W * max(cs.inp.ld, cs.ldraw) * 4 < (1 << 31))

        def wino_codes(cs, which, small=False):
            codes = []
            for base in wino_bases(cs, which):
                codes += [base + c for c in gemm_cands]
                if small:      # small grids - batch-1 inference: also the 8-slot latency ring, as for the direct plans
                    codes += [base + 6418, base + 12818]
            if fused_ok(cs, which):
                codes.append(WINOF)
            return tuple(codes)
        def ws_need(cs):       # split-K scratch of the deepest candidate tried on this shape (x9 small, x4 mid, none big)
            mc = cs.M * max(cs.coutp, cs.cinp)
            need = 9 * mc if mc <= (1 << 21) else (4 * mc if mc <= (1 << 25) else 1)
            # ... and what the library's own heuristic (plan 0: the reference launch of verify-after-tune) asks for: it may
            # split K on a shape the candidate list does not (batch 64 at 832 x 832: the 26 x 26 layers, x3)
            need = max(need, _lib.query('ssp_conv_workspace_floats', B, cs.H, cs.W, cs.cinp, cs.cout, cs.k, 0))
            if not cs.first:
                need = max(need, _lib.query('ssp_conv_workspace_floats', B, cs.H, cs.W, cs.coutp, cs.cin, cs.k, 0))
            for base in wino_bases(cs, 'fwd') + wino_bases(cs, 'dgrad'):
                need = max(need, _lib.query('ssp_conv_workspace_floats', B, cs.H, cs.W, cs.cinp, cs.cout, cs.k, base + 6413))
                need = max(need, _lib.query('ssp_conv_workspace_floats', B, cs.H, cs.W, cs.coutp, cs.cin, cs.k, base + 6413))
            return need
        max_ws = max(ws_need(cs) for cs in elig)
        ws = torch.empty(max_ws, **f32)
        # statistics scratch: the finest tiling any candidate uses (Winograd plans: groups of 16 tiles, counts behind the pairs)
        stats = torch.empty(max(((cs.M + 15) // 16 + 16) * (cs.cout * 2 + 1) for cs in elig), **f32) if which == 'fwd' else None
        gscratch = torch.zeros(max(cs.M * max(cs.inp.ld, cs.ldraw) for cs in elig), **f32) if which == 'dgrad' else None
        verify = os.environ.get('SSP_TUNE_VERIFY', '1') != '0'
        gen = torch.Generator(device=self.device)
        gen.manual_seed(1234)

        def best_of(launch, mn, key, extra=(), direct=None):
            keep = True
            if key in _TUNE_CACHE:       # the same launch shape was timed before (another plan, another model)
                if not wino_tile(_TUNE_CACHE[key]) or _TUNE_CACHE[key] in extra:
                    return _TUNE_CACHE[key]
                keep = False             # a cached Winograd choice whose family is switched off in this process: time the
                                         # other plans, leave the cache entry alone
            best, best_t = 0, None
            fams = {}                    # family (0 direct, 2 / 4 Winograd tile) -> (fastest code, ms)
            # small-batch inference (valid.py runs B = 1): a few dozen tiles cannot stream the filters at HBM speed;
            # deep K splits put every CU on the weight stream
            deep = tuple(bm * 100 + ks * 10 + sl for bm in (64, 128) for ks in (4, 5, 6, 8, 9) for sl in (3, 4, 8)) \
                if mn <= (1 << 21) else ()
            # (direct: the direct-plan candidates of this launch when the standard list does not apply - a thin data
            # gradient, Cin_dx <= 64, ignores plan codes: its heuristic plan alone is timed against the Winograd forms)
            for code in (tuple(direct) if direct is not None else cands + (lat if mn <= (1 << 23) else ()) + deep) + tuple(extra):
                if not wino_tile(code) and ((code // 10) % 10 > 1 or code >= 100000) and mn > (1 << 25):
                    continue      # no split-K scratch for the biggest maps (dozens of waves: nothing to balance)
                try:
                    launch(code)
                    ts = []
                    for _ in range(2):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        launch(code)
                        e1.record()
                        e1.synchronize()
                        ts.append(e0.elapsed_time(e1))
                except _lib.SspError:
                    continue
                t = min(ts)
                if wino_fused(code) and key[0] == 'dgrad':
                    # ONE launch against the three of a through-HBM Winograd plan (transform, GEMM, finishing pass), and in the
                    # step those run next to the filter-gradient stream: measured alone the two forms are within noise of each
                    # other on the 104 x 104 layers (0.50-0.61 against 0.64 ms), in the step the on-chip form is the faster one
                    # (three layers on it: 25.8 ms; one: 26.4 ms - a box whose timings fell the other way).  SSP_ONCHIP_PREFER
                    # (default 0.85) credits it accordingly; 1 = the isolated timing decides.
                    if t >= float(os.environ.get('SSP_ONCHIP_CREDIT_MS', '0.25')):      # (big launches only, as for the filter gradient)
                        t *= float(os.environ.get('SSP_ONCHIP_PREFER', '0.85'))
                if best_t is None or t < best_t * 0.985:     # prefer earlier (simpler) candidates on near-ties
                    best, best_t = code, t
                f_ = wino_tile(code)
                if f_ not in fams or t < fams[f_][1] * 0.985:
                    fams[f_] = (code, t)
            if keep:
                _TUNE_CACHE[key] = best
                _TUNE_FAMILY[key] = fams
            return best

        def admitted(code, key, launch, out_of, operands, bn_of=None, prep=None, primary=True):
            """verify-after-tune (see the docstring): code's result against plan 0's on seeded random operands.  A Winograd
            plan is another ALGORITHM, not another summation order: its bar is 3e-5 of the output's range (measured ~1e-6)."""
            if code == 0 or not verify or _TUNE_VERIFIED.get(key) == code or (key, code) in _TUNE_VERIFIED_ALSO:
                return code
            # Operands are the plan's own buffers (they hold no data yet: tuning runs before the first forward, and
            # Plan.backward's fallback never tunes).  Only the columns a launch owns are randomised: the channel padding of
            # a torch.zeros-allocated activation must stay zero for its consumers.
            for t in operands:
                (t[0].view(-1, t[1])[:, t[2]:t[2] + t[3]] if isinstance(t, tuple) else t).uniform_(-1.0, 1.0, generator=gen)
            if prep is not None:
                prep()                # operands changed: derived operands (Winograd-transformed filters) follow
            bar = 3e-5 if wino_tile(code) else 1e-5
            res = []
            for c in (0, code):
                launch(c)
                r = [out_of().clone()]
                if bn_of is not None:
                    r += bn_of(c)
                res.append(r)
            ok = True
            for a, b in zip(res[0], res[1]):
                den = float(a.abs().max())
                err = float((a - b).abs().max())
                if not (err <= bar * max(den, 1e-30)):      # also false for NaN
                    ok = False
            if ok:
                if primary:
                    _TUNE_VERIFIED[key] = code
                else:
                    _TUNE_VERIFIED_ALSO.add((key, code))
                return code
            TUNE_REJECTED.append((key, code))
            if primary:
                _TUNE_CACHE[key] = 0
            import warnings
            warnings.warn("singleshotpose_amd: verify-after-tune refused plan %d for launch %s (result differs from the "
                          "default plan's); the default plan runs" % (code, key))
            return 0

        for cs in elig:
            if which == 'fwd' and (cs.cout > 64 or fused_ok(cs, which)):
                # operand contents are irrelevant for timing: a channels-last-sized parameter stands in for itself
                wop = cs.conv.weight if cs.cinp == cs.cin else self._wbuf(cs)
                key = ('fwd', B, cs.H, cs.W, cs.cinp, cs.cout, cs.k, cs.inp.ld, cs.ldraw, bool(cs.bn), _tune_tag())

                wc = wino_codes(cs, which, cs.M * cs.coutp <= (1 << 23))

                def prep_f(tile, cs=cs, wop=wop):
                    call('ssp_wino_filter_transform_t', wop.data_ptr(), self._wino_u(cs, tile).data_ptr(), cs.cout, cs.cinp,
                         tile, st)

                def launch(code, cs=cs, wop=wop):
                    wt = self._wino_u(cs, wino_tile(code)) if wino_tile(code) else wop
                    call('ssp_conv_fwd', cs.inp.ptr, wt.data_ptr(), cs.raw.data_ptr(), None,
                         stats.data_ptr() if cs.bn else None, B, cs.H, cs.W, cs.cinp, cs.cout, cs.inp.ld, cs.ldraw,
                         cs.k, 0, code, ws.data_ptr(), max_ws, st)

                def bn_of(code, cs=cs):
                    # batch statistics from this plan's per-tile (mean, M2) partials: mean and 1/std per channel
                    tm = _lib.query('ssp_conv_stats_tile_m', B, cs.H, cs.W, cs.cinp, cs.cout, cs.k, code)
                    nt = _lib.query('ssp_conv_stats_tiles', B, cs.H, cs.W, cs.cinp, cs.cout, cs.k, code)
                    tmp = torch.zeros(8, cs.coutp, **f32)
                    tmp[0].fill_(1.0)
                    tmp[3].fill_(1.0)
                    call('ssp_bn_fwd_finalize', stats.data_ptr(), nt, tm, cs.M, cs.cout,
                         tmp[0].data_ptr(), tmp[1].data_ptr(), tmp[2].data_ptr(), tmp[3].data_ptr(), BN_MOMENTUM,
                         BN_EPS, tmp[4].data_ptr(), tmp[5].data_ptr(), tmp[6].data_ptr(), tmp[7].data_ptr(), st)
                    return [tmp[4, :cs.cout].clone(), tmp[5, :cs.cout].clone()]

                if wc and key not in _TUNE_CACHE:
                    for t in sorted(set(wino_tile(c) for c in wc)):
                        prep_f(t)
                # (Cout <= 64 ignores direct plan codes - its heuristic plan alone is timed against the Winograd forms)
                code = best_of(launch, cs.M * cs.coutp, key, wc, direct=None if cs.cout > 64 else (0,))
                a = cs.inp
                ops_ = [(a.t, a.ld, a.off % a.ld, cs.cinp)] + ([] if wop is cs.conv.weight else [wop])
                cs.plan_fwd = admitted(code, key, launch, lambda cs=cs: cs.raw, ops_, bn_of if cs.bn else None,
                                       (lambda code=code: prep_f(wino_tile(code))) if wino_tile(code) else None)
                # The fastest code of every OTHER family that was timed (direct / F(2x2) / F(4x4)), verified the same way: what
                # the error budget of the forward plans may fall back to (_apply_head_budget); a refused one is dropped.
                cs.fwd_fams = {}
                for f_, (c_, t_) in (_TUNE_FAMILY.get(key) or {}).items():
                    if c_ == cs.plan_fwd:
                        cs.fwd_fams[f_] = (c_, t_)
                    elif c_ != code and admitted(c_, key, launch, lambda cs=cs: cs.raw, ops_, bn_of if cs.bn else None,
                                                 (lambda c_=c_: prep_f(wino_tile(c_))) if wino_tile(c_) else None,
                                                 primary=False) == c_:
                        cs.fwd_fams[f_] = (c_, t_)
                # a direct family whose tuned code was refused still has the library's heuristic plan (code 0, the reference of
                # every verification): the error budget must always be able to fall back to a direct launch
                fams_ = _TUNE_FAMILY.get(key) or {}
                if 0 not in cs.fwd_fams and 0 in fams_:
                    cs.fwd_fams[0] = (0, fams_[0][1])
                # not chosen: the transformed-filter buffers go back to the allocator
                cs.wino_u = {t: b for t, b in (getattr(cs, 'wino_u', None) or {}).items() if t == wino_tile(cs.plan_fwd)}
            if which == 'dgrad' and not cs.first and cs.coutp % 16 == 0 and (cs.cin > 64 or wino_codes(cs, which)):
                key = self._dgrad_key(cs)
                wslice = self._dpack[cs.doff:cs.doff + cs.cinp * cs.k * cs.k * cs.coutp]

                wc = wino_codes(cs, which)

                def prep_d(tile, cs=cs):
                    call('ssp_wino_filter_transform_t', _ptr(self._dpack, cs.doff), self._wino_ud(cs, tile).data_ptr(), cs.cin,
                         cs.coutp, tile, st)

                def launch(code, cs=cs):
                    wt = self._wino_ud(cs, wino_tile(code)).data_ptr() if wino_tile(code) else _ptr(self._dpack, cs.doff)
                    call('ssp_conv_dgrad', cs.raw.data_ptr(), wt, gscratch.data_ptr(), B, cs.H,
                         cs.W, cs.coutp, cs.cin, cs.ldraw, cs.inp.ld, cs.k, 0, code, ws.data_ptr(), max_ws, st)

                if wc and key not in _TUNE_CACHE:
                    for t in sorted(set(wino_tile(c) for c in wc)):
                        prep_d(t)
                code = best_of(launch, cs.M * cs.cinp, key, wc, direct=None if cs.cin > 64 else (0,))
                cs.plan_dgrad = admitted(code, key, launch, lambda cs=cs: gscratch[:cs.M * cs.inp.ld],
                                         [(cs.raw, cs.ldraw, 0, cs.cout), wslice], None,
                                         (lambda code=code: prep_d(wino_tile(code))) if wino_tile(code) else None)
                cs.wino_ud = {t: b for t, b in (getattr(cs, 'wino_ud', None) or {}).items() if t == wino_tile(cs.plan_dgrad)}
        torch.cuda.synchronize()
        if len(_TUNE_CACHE) != n_known:
            _tune_cache_save()
        # the timing / verification scratch (workspaces for the deepest candidate, gradient scratch, twin outputs) goes back
        # to the driver: left in the caching allocator it sits reserved next to the plan's own buffers (multi-scale soak)
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ forward
    def forward(self, x, training, need_grad=False, inline_repack=False):
        B, H, W = self.B, self.H, self.W
        st = torch.cuda.current_stream().cuda_stream
        call = _lib.call
        if x.dtype == torch.uint8:     # (B,H,W,C) image bytes: ToTensor's /255 and the NHWC padding in one pass
            call('ssp_u8hwc_to_nhwc', x.data_ptr(), self.x_nhwc.data_ptr(), B, H, W, self.in_c, self.in_cp, self.in_cp, st)
        else:
            call('ssp_nchw_to_nhwc', x.data_ptr(), self.x_nhwc.data_ptr(), B, self.in_c, H, W, self.in_cp, self.in_cp, st)
        if training and need_grad and self._head_budget_done and self._tune:
            every = int(os.environ.get('SSP_HEAD_BUDGET_EVERY', '256'))
            if every > 0 and self.generation - self._hb_checked >= every:
                self._hb_checked = self.generation
                drifted = self.head_budget_drifted()
                sync = getattr(self.net, '_plan_sync', None)      # multi-GPU: BatchNorm statistics are per replica, the plan set
                if sync is not None and hasattr(sync, 'all_ok'):  # is not - every rank re-measures when any rank drifted
                    drifted = not sync.all_ok(not drifted)
                if drifted:
                    self.head_budget_stale()
        if training and need_grad and not self._head_budget_done:
            self._apply_head_budget()        # on the plan's first training batch, after load_weights, and when the BatchNorm
                                             # gains drifted (network-level rounding budget)
            self._sync_codes(0)              # multi-GPU: every rank runs rank 0's forward codes
        # (The whole training step as two captured hipGraphs was built and measured in round 4 - profiles/r04_step_graph.txt:
        # a replayed two-stream chain of ~300 nodes is SLOWER than launching it, 9.1 ms against 6.0 ms at batch 8 - and
        # removed in round 5: it mirrored this method's host-side state by hand.  Inference keeps its graph, forward_graph.)
        self._forward_body(training, need_grad, inline_repack)
        if training and getattr(self, '_hb_gains_after_forward', False):
            self._hb_gains_after_forward = False
            self._hb_gains = self._bn_gains()
        o = self.out_act
        y = torch.empty(B, o.C, o.H, o.W, dtype=torch.float32, device=self.device)
        call('ssp_nhwc_to_nchw', o.ptr, y.data_ptr(), B, o.C, o.H, o.W, o.ld, st)
        self.consumed = False
        self.was_training = training
        self.generation += 1
        return y

    def _forward_body(self, training, need_grad, inline_repack):
        B, H, W = self.B, self.H, self.W
        st = torch.cuda.current_stream().cuda_stream
        call = _lib.call
        # Filter repacks depend only on the weights, not on the activations: they run on the side stream, ahead of the
        # convolutions that use them, instead of as one more dependent launch
        # in front of every conv on the main stream.  Forward operands first (two events: the first four layers, then
        # the rest), then - when a backward will follow - the flipped/transposed data-gradient operands, which hide
        # under the MFMA-bound forward convs instead of sitting on the critical path of backward.
        # eval: repack only when a parameter changed (in-place updates bump _version, the fused SGD bumps the weights
        # epoch, load_weights invalidates explicitly); training: weights change every step, always repack.
        # Parameters stored channels-last (Darknet builds them that way) ARE the forward operand: nothing to repack.
        stale = []
        for op in self.ops_fwd:
            if op[0] == 'conv':
                cs = op[1]
                wt = cs.conv.weight
                cs.packed = _is_packed(wt, cs.cinp)
                if cs.packed:
                    continue
                key = (wt.data_ptr(), wt._version, _WEIGHTS_EPOCH[0])
                if inline_repack:
                    # graph capture: the repack is part of the captured chain (stays on this stream, runs every replay)
                    call('ssp_repack_fwd', wt.detach().contiguous().data_ptr(), self._wbuf(cs).data_ptr(), cs.cout,
                         cs.cin, cs.cinp, cs.k, st)
                    self.wversion.pop(cs.ind, None)
                elif training or self.wversion.get(cs.ind) != key:
                    stale.append((cs, key))
        if need_grad:
            self._prepare_backward()
        # Winograd-plan layers read TRANSFORMED filters: re-derived when the weights may have changed (every training
        # step; in eval when a parameter version / the weights epoch moved), on the side stream like the repacks
        wino = []
        for cs in self.convs.values():
            if wino_tile(cs.plan_fwd):
                wt = cs.conv.weight
                wkey = (wt.data_ptr(), wt._version, _WEIGHTS_EPOCH[0])
                if inline_repack or training or self.wino_version.get(cs.ind) != wkey:
                    wino.append((cs, wkey))
        wait_for = {}
        if inline_repack:
            for cs, _ in wino:       # graph capture: part of the captured chain
                src = cs.conv.weight if cs.packed else self._wbuf(cs)
                tile = wino_tile(cs.plan_fwd)
                call('ssp_wino_filter_transform_t', src.data_ptr(), self._wino_u(cs, tile).data_ptr(), cs.cout, cs.cinp, tile, st)
                self.wino_version.pop(cs.ind, None)
            wino = []
        if stale or need_grad or wino:
            if self.side_stream is None:
                self.side_stream = _side_stream(self.device)
            side = self.side_stream
            side.wait_stream(torch.cuda.current_stream())
            for group in (stale[:4], stale[4:]):
                for cs, key in group:
                    with torch.cuda.stream(side):
                        src = cs.conv.weight.detach().contiguous()      # the repack kernels read (Cout,Cin,kh,kw) order
                        src.record_stream(side)
                    call('ssp_repack_fwd', src.data_ptr(), self._wbuf(cs).data_ptr(), cs.cout, cs.cin,
                         cs.cinp, cs.k, side.cuda_stream)
                    self.wversion[cs.ind] = key
                if group:
                    ev = side.record_event()
                    for cs, _ in group:
                        wait_for[cs.ind] = ev
        if stale or need_grad or wino:
            # the forward filter transforms of every Winograd layer, queued up front: they depend on the weights only and
            # run under the first block and layer 2.  (Queueing them layer by layer just ahead of their convs, and the
            # data-gradient operands after the last forward launch, was tried for the host-bound look of a traced batch-8
            # step - profiles/r04_timeline_b8.txt: first kernel 0.65 ms late - and bought nothing: the host queues a step in
            # 3.6 ms, tools/host_issue_time.py, against 6 ms of GPU time at batch 8 and 28 ms at batch 64 - it runs ahead of the
            # GPU either way and the step time did not move.)
            if wino:
                for cs, wkey in wino:
                    src = cs.conv.weight if cs.packed else self._wbuf(cs)
                    tile = wino_tile(cs.plan_fwd)
                    call('ssp_wino_filter_transform_t', src.data_ptr(), self._wino_u(cs, tile).data_ptr(), cs.cout, cs.cinp,
                         tile, side.cuda_stream)
                    self.wino_version[cs.ind] = wkey
                # ONE event behind all of them: the side stream is through by ~1.4 ms, the main stream reaches layer 4 at
                # ~1.6 ms, and every cross-stream wait costs ~17 us of GPU idle even when its event has long fired (one
                # event per layer: 13 such gaps, forward idle 0.68 ms instead of 0.3-0.6, profiles/r04_timeline.txt)
                ev = side.record_event()
                for cs, _ in wino:
                    wait_for[cs.ind] = ev
        if need_grad:
            # (queueing these BEHIND the last on-chip Winograd forward launch - that kernel is persistent, one workgroup per CU,
            # and shares the chip badly - was measured: 26.17 / 25.93 ms against 25.85 / 25.97 ms as is, nothing)
            for ind in sorted(self.convs.keys(), reverse=True):
                cs = self.convs[ind]
                if not cs.first:
                    self._repack_dgrad(cs, side)
            self.dgrad_ready = side.record_event()
        else:
            self.dgrad_ready = None
        waited = set()
        if training:
            self.net._bn_epoch += 1       # running statistics change below: every plan's inference constants are stale
        for op in self.ops_fwd:
            kind = op[0]
            if kind == 'conv':
                cs = op[1]
                ev = wait_for.get(cs.ind)
                if ev is not None and id(ev) not in waited:
                    torch.cuda.current_stream().wait_event(ev)      # this layer's packed filters are ready
                    waited.add(id(ev))
                bias = cs.conv.bias.data_ptr() if cs.conv.bias is not None else None
                use_stats = cs.bn and training
                v = cs.vec
                if cs.bn and not training:
                    # inference-mode BatchNorm is a per-channel affine map of constants: recomputed only when one of
                    # its four tensors changed (in-place updates bump _version; load_weights / fused SGD bump the epoch)
                    bn = cs.bnm
                    # (a training-mode forward rewrites running_mean / running_var through raw pointers and reuses the
                    # scale / shift vectors for the batch statistics: it bumps the net's BN epoch, also part of the key)
                    bkey = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean,
                                                                     bn.running_var)) + (_WEIGHTS_EPOCH[0],
                                                                                         self.net._bn_epoch)
                    if inline_repack or self.bnversion.get(cs.ind) != bkey:
                        call('ssp_bn_eval_prepare', cs.cout, bn.weight.data_ptr(), bn.bias.data_ptr(),
                             bn.running_mean.data_ptr(), bn.running_var.data_ptr(), BN_EPS, v[0].data_ptr(),
                             v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), st)
                        self.bnversion[cs.ind] = None if inline_repack else bkey
                wptr = cs.conv.weight.data_ptr() if cs.packed else self._wbuf(cs).data_ptr()
                if wino_tile(cs.plan_fwd):
                    wptr = self._wino_u(cs, wino_tile(cs.plan_fwd)).data_ptr()
                cs.first_live = False
                if cs.first_fused and not cs.packed and (training or not need_grad):
                    # training: statistics pass + apply pass; inference: the apply pass alone with the running-statistics
                    # affine (v[2], v[3] from ssp_bn_eval_prepare above) - conv + BN + leaky + pool in one launch, the
                    # full-resolution map is never written (672 x 672, batch 1: 12 us instead of 23 + 12)
                    if training:
                        bn = cs.bnm
                        call('ssp_first_fwd_stats', cs.inp.ptr, wptr, cs.stats.data_ptr(), B, cs.H, cs.W, st)
                        call('ssp_bn_fwd_finalize', cs.stats.data_ptr(), cs.first_groups, cs.first_tile, cs.M, cs.cout,
                             bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                             bn.running_var.data_ptr(), self.bn_momentum, BN_EPS, v[0].data_ptr(), v[1].data_ptr(),
                             v[2].data_ptr(), v[3].data_ptr(), st)
                    call('ssp_first_fwd_apply', cs.inp.ptr, wptr, v[2].data_ptr(), v[3].data_ptr(), cs.slope,
                         cs.out.ptr, cs.out.ld, B, cs.H, cs.W, st)
                    cs.first_live = training        # only a training-mode forward can be followed by the fused backward
                    continue
                if not training and not need_grad and cs.needs_act and not cs.pool and cs.coutp == cs.cout:
                    # inference, un-pooled block: BatchNorm affine + leaky folded into the conv epilogue - one launch,
                    # no raw-output round trip (backward needs the raw output, so training / autograd keep two steps)
                    call('ssp_conv_fwd_affine', cs.inp.ptr, wptr, cs.out.ptr, v[2].data_ptr() if cs.bn else None,
                         v[3].data_ptr() if cs.bn else bias, cs.slope, B, cs.H, cs.W, cs.cinp, cs.cout, cs.inp.ld,
                         cs.out.ld, cs.k, cs.plan_fwd, self.ws.data_ptr(), self.ws_floats, st)
                    continue
                ws_t = self.ws
                cs.v_live = False
                share_v = os.environ.get('SSP_WINO_SHARE_V', '1') != '0'
                if (need_grad and wino_tile(cs.plan_fwd) and not wino_fused(cs.plan_fwd) and
                        getattr(cs, 'wgrad_wino', 0) == wino_tile(cs.plan_fwd) and share_v):
                    ws_t = self._wino_ws(cs)         # V stays at the head of this buffer for the layer's filter gradient
                    cs.v_live = True
                elif (need_grad and getattr(cs, 'wgrad_wino', 0) and share_v and self.side_stream is not None and
                        os.environ.get('SSP_WINO_EARLY_V', '0') == '1'):
                    # The filter gradient runs in the Winograd domain but this forward launch does not leave its V behind (the
                    # error budget moved the layer to a direct code, or to the other tile size): the input transform the
                    # filter gradient needs CAN be queued now on the second stream (SSP_WINO_EARLY_V=1) - an HBM-bound pass next
                    # to the forward launches instead of inside the backward pass.  Measured on one box, two interleaved rounds
                    # (profiles/r05_step_ab.txt): 27.48 / 27.57 ms with it, 27.38 / 27.47 without - the transform slows the
                    # MFMA-bound direct launches it runs beside by as much as it saves later; off by default.
                    wws = self._wino_ws(cs)
                    ready = torch.cuda.current_stream().record_event()      # the layer's input is complete on the main stream
                    self.side_stream.wait_event(ready)
                    call('ssp_wino_input_transform_t', cs.inp.ptr, cs.inp.ld, wws.data_ptr(), B, cs.H, cs.W, cs.cinp,
                         cs.wgrad_wino, self.side_stream.cuda_stream)
                    cs.v_live = True
                call('ssp_conv_fwd', cs.inp.ptr, wptr, cs.raw.data_ptr(), bias,
                     cs.stats.data_ptr() if use_stats else None, B, cs.H, cs.W, cs.cinp, cs.cout, cs.inp.ld, cs.ldraw,
                     cs.k, 0, cs.plan_fwd, ws_t.data_ptr(), ws_t.numel(), st)
                if cs.bn and training:
                    bn = cs.bnm
                    call('ssp_bn_fwd_finalize', cs.stats.data_ptr(), cs.ntile, cs.tile_m, cs.M, cs.cout,
                         bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                         bn.running_var.data_ptr(), self.bn_momentum, BN_EPS, v[0].data_ptr(), v[1].data_ptr(),
                         v[2].data_ptr(), v[3].data_ptr(), st)
                if cs.needs_act:
                    call('ssp_bn_act_fwd', cs.raw.data_ptr(), cs.ldraw, cs.out.ptr, cs.out.ld, v[2].data_ptr(),
                         v[3].data_ptr(), cs.coutp, B, cs.H, cs.W, 1 if cs.pool else 0, cs.slope, st)
            elif kind == 'maxpool':
                _, ind, src, out = op
                call('ssp_maxpool_fwd', src.ptr, src.ld, out.ptr, out.ld, _pad4(src.C), B, src.H, src.W, st)
            elif kind == 'reorg':
                _, ind, src, out = op
                call('ssp_reorg', src.ptr, src.ld, out.ptr, out.ld, src.C, B, src.H, src.W, 0, 0, st)
            elif kind == 'concat':
                _, ind, layers, srcs, out = op
                off = 0
                for s in srcs:
                    call('ssp_copy_channels', s.ptr, s.ld, _ptr(out.t, off), out.ld, s.C, B * s.H * s.W, 0, st)
                    off += s.C

    # ------------------------------------------------------------------ inference as one hipGraph
    def _graph_tensors(self):
        ts = []
        for cs in self.convs.values():
            ts.append(cs.conv.weight)
            if cs.conv.bias is not None:
                ts.append(cs.conv.bias)
            if cs.bn:
                ts += [cs.bnm.weight, cs.bnm.bias, cs.bnm.running_mean, cs.bnm.running_var]
        return ts

    def forward_graph(self, x):
        """Eval-mode forward replayed from a captured hipGraph: the ~75 launches of a forward pass are launch-bound at
        small batch (valid.py runs B = 1: 1.8 ms eager for 0.6 ms of kernels).  The kernels read parameters and BN
        running statistics from their own memory, so in-place weight updates need no re-capture; a parameter that moved
        (new data_ptr), another input dtype, or a non-contiguous first-layer filter does."""
        if self._graph_failed:
            return self.forward(x, False)
        key = (x.dtype,) + tuple((t.data_ptr(), tuple(t.stride())) for t in self._graph_tensors())
        if self._graph is None or self._graph_key != key:
            try:
                self._x_static = torch.empty_like(x)
                self._x_static.copy_(x)
                self.forward(self._x_static, False, inline_repack=True)      # eager warm-up (lazy per-kernel set-up)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    y = self.forward(self._x_static, False, inline_repack=True)
                self._graph, self._y_static, self._graph_key = g, y, key
            except Exception as e:      # capture unsupported in this runtime: keep the eager HIP launches
                import warnings
                warnings.warn("hipGraph capture of the inference chain failed (%s); using eager launches" % (e,))
                self._graph, self._graph_failed = None, True
                torch.cuda.synchronize(self.device)
                return self.forward(x, False)
        self._x_static.copy_(x)
        self._graph.replay()
        self.consumed = True          # no backward on a graph replay
        return self._y_static.clone()

    # ------------------------------------------------------------------ backward
    def _grad_buf(self, ind, like):
        g = self.grads.get(ind)
        if g is None:
            g = _Act(torch.empty(self.B * like.H * like.W * like.ld, dtype=torch.float32, device=self.device), 0,
                     like.C, like.H, like.W, like.ld)
            self.grads[ind] = g
        return g

    def _dgrad(self, cs, src, dy_ptr, dy_ld, written, fused_stats, st):
        """Data gradient of conv block `cs` into the gradient buffer of its input's producer `src` (with the producer's
        BatchNorm-backward sums folded in where _plan_bn_fusion arranged it)."""
        call = _lib.call
        B = self.B
        gin = self._grad_buf(src, cs.inp)
        scs = getattr(cs, 'bn_fuse_src', None)
        dwt = (self._wino_ud(cs, wino_tile(cs.plan_dgrad)).data_ptr() if wino_tile(cs.plan_dgrad)
               else _ptr(self._dpack, cs.doff))
        if scs is not None and src not in written:
            sv = scs.vec
            call('ssp_conv_dgrad_bnbwd', dy_ptr, dwt, gin.ptr, B, cs.H, cs.W,
                 cs.coutp, cs.cin, dy_ld, gin.ld, cs.k, cs.plan_dgrad, self.ws.data_ptr(), self.ws_floats,
                 scs.raw.data_ptr(), scs.ldraw, sv[2].data_ptr(), sv[3].data_ptr(), sv[0].data_ptr(),
                 sv[1].data_ptr(), scs.slope, scs.bnp[0].data_ptr(), scs.bnp[1], st)
            fused_stats.add(src)
        else:
            call('ssp_conv_dgrad', dy_ptr, dwt, gin.ptr, B, cs.H, cs.W, cs.coutp,
                 cs.cin, dy_ld, gin.ld, cs.k, 1 if src in written else 0, cs.plan_dgrad, self.ws.data_ptr(),
                 self.ws_floats, st)
        written.add(src)

    def backward(self, grad_out):
        """grad_out: (B, C, h, w) NCHW.  Returns {param tensor id: grad tensor}."""
        if self.consumed:
            raise RuntimeError("Darknet backward called twice on the same forward: the HIP path rewrites the saved "
                               "conv outputs in place (no retain_graph support)")
        self.consumed = True
        B = self.B
        st = torch.cuda.current_stream().cuda_stream
        call = _lib.call
        o = self.out_act
        g_last = self._grad_buf(self.last, o)
        call('ssp_nchw_to_nhwc', grad_out.data_ptr(), g_last.ptr, B, o.C, o.H, o.W, o.C, o.ld, st)
        if o.ld > o.C:
            raise NotImplementedError("network output channels must be a multiple of 4")
        if self.side_stream is None:
            self.side_stream = _side_stream(self.device)
        # ONE flat gradient buffer per model and device, reused by every backward of every plan (the layout depends on the
        # model only): the returned gradients are views of it.  Reuse is safe only while nothing else still views the
        # buffer (optimizer.zero_grad(set_to_none=True) - torch's default - drops the .grad views; a caller that keeps or
        # accumulates gradients across backwards gets a fresh buffer, the old one stays with the tensors that view it).
        # Zeroed (in the body): the filter-gradient kernel accumulates channels-last parameters' gradients straight into it.
        ent = self.net._flat_grads.get(self.device)
        flat = None
        if ent is not None and ent[0].numel() == self.grad_total and not _storage_shared(ent[0], ent[1], self.net._params()):
            flat = ent[0]
        if flat is None:
            flat = torch.empty(self.grad_total, dtype=torch.float32, device=self.device)
            self.net._flat_grads[self.device] = (flat, _storage_refs(flat))
        flat.record_stream(self.side_stream)
        self.last_flat_grad = flat
        return self._backward_body(flat)

    def _backward_body(self, flat):
        B = self.B
        st = torch.cuda.current_stream().cuda_stream
        call = _lib.call
        blocks = self.net.blocks
        written = set()
        fused_stats = set()      # blocks whose BatchNorm-backward reductions were produced by their consumer's dgrad launch
        written.add(self.last)
        flat.zero_()
        for cs in self.convs.values():          # gradient staging of the parameters that are not channels-last
            if not cs.packed and not cs.first_live:      # (the fused first block's filter gradient is WRITTEN, not accumulated)
                self._gbuf(cs).zero_()
        # Filter gradients run on a second stream: wgrad(l) only needs dY(l) and the saved input activation, so it
        # overlaps the dgrad(l) -> BN-backward(l-1) chain of the main stream and fills the idle CUs of its last wave.
        main = torch.cuda.current_stream()
        side = main if self.serial_backward else self.side_stream
        st2 = side.cuda_stream
        if self.dgrad_ready is not None:
            main.wait_event(self.dgrad_ready)       # dgrad filter repacks were queued during forward
        else:
            self._prepare_backward(tune=False)      # the saved conv outputs are live: no timing / verify launches now
            for ind in sorted(self.convs.keys(), reverse=True):   # forward ran without grad bookkeeping: repack now
                cs = self.convs[ind]
                if not cs.first:
                    self._repack_dgrad(cs, main)
        out_grads = {}
        training = self.was_training
        tail_sched = side is not main and os.environ.get('SSP_TAIL_SCHED', '1') != '0'
        if self.reducer is not None:
            self.reducer.begin(flat)

        def gview(prm, channels_last=False):
            off, n, shape = self.grad_layout[id(prm)]
            if channels_last:      # the parameter's own strides: [Cout][kh][kw][Cin] in memory
                return torch.as_strided(flat, shape, prm.stride(), off)
            return flat[off:off + n].view(shape)

        def producer_of(act):
            for i, a in enumerate(self.acts):
                if a is act:
                    return i
            return None

        for ind in range(self.nl - 1, -1, -1):
            block = blocks[ind + 1]
            t = block['type']
            if t in ('region', 'cost'):
                continue
            if t == 'maxpool' and ind in self.fused_pool:
                continue  # handled by the conv block that owns it
            if t == 'convolutional':
                cs = self.convs[ind]
                oind = ind + 1 if cs.pool else ind
                if oind not in written:
                    continue
                g = self.grads[oind]
                v = cs.vec
                if cs.first_live:
                    # first block, fused form: both backward passes recompute the convolution from the input
                    dgam, dbet = gview(cs.bnm.weight), gview(cs.bnm.bias)
                    out_grads[id(cs.bnm.weight)], out_grads[id(cs.bnm.bias)] = dgam, dbet
                    wptr = self._wbuf(cs).data_ptr()
                    call('ssp_first_bwd_reduce', cs.inp.ptr, wptr, g.ptr, g.ld, v[2].data_ptr(), v[3].data_ptr(),
                         v[0].data_ptr(), v[1].data_ptr(), cs.slope, cs.first_partial.data_ptr(), B, cs.H, cs.W, st)
                    call('ssp_bn_bwd_finalize', cs.first_partial.data_ptr(), cs.first_groups, cs.cout, cs.M,
                         1 if training else 0, 0, dgam.data_ptr(), dbet.data_ptr(), v[4].data_ptr(), v[5].data_ptr(), st)
                    # The step's tail is a dependency chain: dgrad of the block's consumer -> this reduce -> this filter
                    # gradient (HBM-bound: it re-reads the 1.4 GB output gradient).  With the tail schedule the consumer's
                    # filter gradient (MFMA-bound) was held back behind its data gradient and is running on the side
                    # stream NOW: this pass stays on the main stream and overlaps it, instead of queueing behind it.
                    fst = st if tail_sched else st2
                    if not tail_sched:
                        side.wait_stream(main)
                    gw = gview(cs.conv.weight, False)
                    if getattr(cs, 'first_wpart', None) is None:      # per-workgroup partial gradients, summed in float64
                        cs.first_wpart = torch.empty(_lib.query('ssp_first_wgrad_workspace_floats', B, cs.H, cs.W),
                                                     dtype=torch.float32, device=self.device)
                    call('ssp_first_bwd_wgrad', cs.inp.ptr, wptr, g.ptr, g.ld, v[2].data_ptr(), v[3].data_ptr(),
                         v[0].data_ptr(), v[1].data_ptr(), v[4].data_ptr(), v[5].data_ptr(), cs.slope,
                         self._gbuf(cs).data_ptr(), cs.first_wpart.data_ptr(), cs.first_wpart.numel(), B, cs.H, cs.W, fst)
                    call('ssp_unpack_grad', self._gbuf(cs).data_ptr(), gw.data_ptr(), cs.cout, cs.cin, cs.cinp, cs.k, fst)
                    out_grads[id(cs.conv.weight)] = gw
                    if self.reducer is not None:
                        if tail_sched:
                            side.wait_stream(main)
                        with torch.cuda.stream(side):
                            self.reducer.layer_done(flat, cs.grad_lo, cs.grad_hi)
                    continue
                if cs.needs_act:
                    if cs.bn and cs.coutp == cs.cout:
                        dgam, dbet = gview(cs.bnm.weight), gview(cs.bnm.bias)
                        out_grads[id(cs.bnm.weight)], out_grads[id(cs.bnm.bias)] = dgam, dbet
                        dg_ptr, db_ptr = dgam.data_ptr(), dbet.data_ptr()
                        # two-level reduction (per-workgroup partials -> fp64 finalize).  The single-pass form of
                        # ssp_bn_act_bwd (atomics into the zeroed gradient, no finalize launch) measured 1.1 ms SLOWER
                        # per step: 1024 workgroups hammering the same 2*C addresses serialise in the L2.
                        partial = self.bn_partial.data_ptr()
                    else:
                        dg_ptr, db_ptr = v[6].data_ptr(), v[7].data_ptr()
                        partial = self.bn_partial.data_ptr()
                    if cs.ind in fused_stats:
                        # the two reductions came out of the consumer's data-gradient launch: finalize + apply only
                        ptile, rows, folded = cs.bnp
                        call('ssp_bn_act_bwd_partials', cs.raw.data_ptr(), cs.ldraw, g.ptr, g.ld, cs.raw.data_ptr(),
                             cs.ldraw, v[2].data_ptr(), v[3].data_ptr(), v[0].data_ptr(), v[1].data_ptr(), cs.coutp, B,
                             cs.H, cs.W, cs.slope, 1 if training else 0, ptile.data_ptr(), rows, 1 if folded else 0,
                             dg_ptr, db_ptr, v[4].data_ptr(), v[5].data_ptr(), st)
                    else:
                        call('ssp_bn_act_bwd', cs.raw.data_ptr(), cs.ldraw, g.ptr, g.ld, cs.raw.data_ptr(), cs.ldraw,
                             v[2].data_ptr(), v[3].data_ptr(), v[0].data_ptr(), v[1].data_ptr(), cs.coutp, B, cs.H, cs.W,
                             1 if cs.pool else 0, cs.slope, 1 if (training and cs.bn) else 0, partial,
                             dg_ptr, db_ptr, v[4].data_ptr(), v[5].data_ptr(), st)
                    dy_ptr, dy_ld = cs.raw.data_ptr(), cs.ldraw
                    if cs.bn and cs.coutp != cs.cout:
                        gview(cs.bnm.weight).copy_(v[6][:cs.cout])
                        gview(cs.bnm.bias).copy_(v[7][:cs.cout])
                        out_grads[id(cs.bnm.weight)], out_grads[id(cs.bnm.bias)] = gview(cs.bnm.weight), gview(cs.bnm.bias)
                else:
                    dy_ptr, dy_ld = g.ptr, g.ld
                if cs.conv.bias is not None:
                    db = gview(cs.conv.bias)
                    call('ssp_colsum', dy_ptr, dy_ld, cs.M, cs.cout, db.data_ptr(), st)
                    out_grads[id(cs.conv.bias)] = db
                src = None if cs.first else producer_of(cs.inp)
                # tail schedule: the consumer of the fused first block (layer 2) runs its data gradient - the head of the
                # chain that ends the step - BEFORE its filter gradient is released on the side stream
                owner = None if src is None else (self.convs.get(src - 1) if src in self.fused_pool else self.convs.get(src))
                defer = tail_sched and owner is not None and owner.first_live
                if defer:
                    self._dgrad(cs, src, dy_ptr, dy_ld, written, fused_stats, st)
                side.wait_stream(main)          # dY(l) (and the zeroed packed-gradient buffer) are ready
                gw = gview(cs.conv.weight, cs.packed)
                if cs.packed and getattr(cs, 'wgrad_wino', 0):
                    wws = self._wino_ws(cs)
                    call('ssp_conv_wgrad_wino_t', dy_ptr, None if getattr(cs, 'v_live', False) else cs.inp.ptr, gw.data_ptr(),
                         B, cs.H, cs.W, cs.cinp, cs.cout, dy_ld, cs.inp.ld, cs.wgrad_wino, wws.data_ptr(), wws.numel(), st2)
                    cs.v_live = False
                elif cs.packed:       # accumulate in place: the gradient has the parameter's channels-last layout
                    call('ssp_conv_wgrad', dy_ptr, cs.inp.ptr, gw.data_ptr(), B, cs.H, cs.W, cs.cinp, cs.cout, dy_ld,
                         cs.inp.ld, cs.k, st2)
                else:
                    call('ssp_conv_wgrad', dy_ptr, cs.inp.ptr, self._gbuf(cs).data_ptr(), B, cs.H, cs.W, cs.cinp,
                         cs.cout, dy_ld, cs.inp.ld, cs.k, st2)
                    call('ssp_unpack_grad', self._gbuf(cs).data_ptr(), gw.data_ptr(), cs.cout, cs.cin, cs.cinp, cs.k, st2)
                out_grads[id(cs.conv.weight)] = gw
                if self.reducer is not None:
                    with torch.cuda.stream(side):   # the all-reduce of a finished bucket is ordered after its wgrads
                        self.reducer.layer_done(flat, cs.grad_lo, cs.grad_hi)
                if not cs.first and not defer:
                    self._dgrad(cs, src, dy_ptr, dy_ld, written, fused_stats, st)
            elif t == 'maxpool':
                if ind not in written:
                    continue
                op = [o_ for o_ in self.ops_fwd if o_[0] == 'maxpool' and o_[1] == ind][0]
                _, _, src, out = op
                sind = producer_of(src)
                gin = self._grad_buf(sind, src)
                g = self.grads[ind]
                call('ssp_maxpool_bwd', src.ptr, src.ld, g.ptr, g.ld, gin.ptr, gin.ld, _pad4(src.C), B, src.H, src.W,
                     1 if sind in written else 0, st)
                written.add(sind)
            elif t == 'reorg':
                if ind not in written:
                    continue
                op = [o_ for o_ in self.ops_fwd if o_[0] == 'reorg' and o_[1] == ind][0]
                _, _, src, out = op
                sind = producer_of(src)
                gin = self._grad_buf(sind, src)
                g = self.grads[ind]
                call('ssp_reorg', g.ptr, g.ld, gin.ptr, gin.ld, src.C, B, src.H, src.W, 1,
                     1 if sind in written else 0, st)
                written.add(sind)
            elif t == 'route':
                if ind not in written:
                    continue
                layers = resolve_layers(block['layers'], ind)
                g = self.grads[ind]
                off = 0
                for l in layers:
                    a = self.acts[l]
                    src = producer_of(a)
                    gin = self._grad_buf(src, a)
                    if gin is not g:
                        call('ssp_copy_channels', _ptr(g.t, g.off + off), g.ld, gin.ptr, gin.ld, a.C, B * a.H * a.W,
                             1 if src in written else 0, st)
                    written.add(src)
                    off += a.C
        main.wait_stream(side)                  # every filter gradient is complete before autograd hands them out
        return out_grads


class _DarknetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are sequences of HIP launches."""

    @staticmethod
    def forward(ctx, plan, training, x, *params):
        ctx.plan = plan
        ctx.params = params
        y = plan.forward(x, training, need_grad=True)
        ctx.generation = plan.generation
        return y

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan if ctx.plan is not None else ctx.plan_ref()
        if plan is None:
            raise RuntimeError("Darknet backward called twice on the same forward, and the plan of that input shape has "
                               "left the plan cache since (another resolution took its memory): run the forward again")
        if plan.generation != ctx.generation:
            raise RuntimeError("Darknet backward after a newer forward on the same input shape: the plan's saved "
                               "activations were overwritten")
        grads = plan.backward(grad_out.contiguous())
        # Until its backward has run the node keeps its plan alive (forward at shape A, forward at shape B, backward of A
        # works whatever the cache evicted).  Afterwards only the cache does: a caller that still holds the loss tensor of the
        # previous resolution (every training loop does, until it assigns the next one) must not keep that resolution's
        # 30 - 65 GB of buffers next to the new plan's - that pair was the peak of the multi-scale soak.
        ctx.plan_ref = weakref.ref(plan)
        ctx.plan = None
        res = [None, None, None]
        for p in ctx.params:
            res.append(grads.get(id(p)))
        return tuple(res)


class _DarknetEvalFn(torch.autograd.Function):
    """Inference-mode forward with autograd enabled - what the reference's unchanged valid.py / train.py test() run:
    `Variable(data, volatile=True)` (valid.py:113) is a no-op on current torch, so grad mode stays on.  The forward takes
    the cheap inference chain (fused conv + BN affine + leaky launches, cached BN constants, no data-gradient operand
    repacks); in the rare case that somebody does call backward on it (frozen-BN fine-tuning), backward first re-runs
    the forward in its activation-keeping form and then the normal backward chain."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        ctx.plan = plan
        ctx.params = params
        ctx.x = x
        ctx.xver = x._version
        return plan.forward(x, False)

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan
        if ctx.x._version != ctx.xver:
            raise RuntimeError("Darknet (eval mode) backward: the input tensor was modified in place after the forward")
        plan.forward(ctx.x, False, need_grad=True)      # recompute, keeping the raw conv outputs
        grads = plan.backward(grad_out.contiguous())
        res = [None, None]
        for p in ctx.params:
            res.append(grads.get(id(p)))
        return tuple(res)
