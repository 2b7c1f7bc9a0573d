"""Darknet(nn.Module) on hand-written HIP kernels - host-side mirror of the reference's darknet.py.

Same surface as /root/reference/darknet.py:59-394: Darknet(cfgfile), forward(x NCHW fp32) -> raw head
(B, nA*(2K+1+nC), H/32, W/32), print_network, load_weights, load_weights_until_last, save_weights, attributes
width/height/test_width/test_height/num_keypoints/anchors/num_anchors/anchor_step/num_classes/seen/iter/header/
blocks/models/loss.  `models` keeps the reference's module tree (models[i][0] = conv, models[i][1] = bn,
parameter names models.N.conv{k}.weight / models.N.bn{k}.*) so checkpoints, optimisers and
`named_parameters()` filters see the same thing; the nn.Conv2d / nn.BatchNorm2d objects are parameter holders only -
forward never calls them.  All device work goes through libssp_hip.so (engine.Plan); there is no CPU or eager
fallback: a CPU tensor or a missing library raises.
"""
import collections
import gc
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .cfg import (load_conv, load_conv_bn, load_fc, parse_cfg, print_cfg, resolve_layers, save_conv, save_conv_bn,
                  save_fc)
from .engine import Plan, _DarknetEvalFn, _DarknetFn, weights_changed
from .region_loss import RegionLoss, RegionLossMulti


class _HipOnly(nn.Module):
    """Structural placeholder: the op runs inside the fused HIP plan, never as a standalone torch module."""

    def forward(self, x):
        raise RuntimeError("%s is executed by the HIP plan of Darknet.forward; it has no standalone eager path"
                           % type(self).__name__)


class MaxPoolStride1(_HipOnly):
    pass


class Reorg(_HipOnly):
    def __init__(self, stride=2):
        super(Reorg, self).__init__()
        self.stride = stride


class GlobalAvgPool2d(_HipOnly):
    pass


class EmptyModule(nn.Module):
    def forward(self, x):
        return x


class Darknet(nn.Module):
    _loss_cls = RegionLoss

    def __init__(self, cfgfile):
        super(Darknet, self).__init__()
        self.blocks = parse_cfg(cfgfile)
        self.models = self.create_network(self.blocks)
        self.loss = self.models[len(self.models) - 1]

        net = self.blocks[0]
        self.width = int(net['width'])
        self.height = int(net['height'])
        if 'test_width' in net:   # the multi-object cfg has no test size / num_keypoints (darknet_multi.py:66-70)
            self.test_width = int(net['test_width'])
            self.test_height = int(net['test_height'])
        if 'num_keypoints' in net:
            self.num_keypoints = int(net['num_keypoints'])

        if self.blocks[-1]['type'] == 'region':
            self.anchors = self.loss.anchors
            self.num_anchors = self.loss.num_anchors
            self.anchor_step = self.loss.anchor_step
            self.num_classes = self.loss.num_classes

        self.header = torch.IntTensor([0, 0, 0, 0])
        self.seen = 0
        self.iter = 0
        self._plans = collections.OrderedDict()
        self._flat_grads = {}       # device -> flat gradient buffer shared by every plan's backward (engine.Plan.backward)
        self._bn_epoch = 0          # bumped by every training-mode forward (engine.Plan: inference BN constants key)
        self._max_plans = 32
        # eval forward as one captured hipGraph replay (Plan.forward_graph).  Opt-in: measured no gain on MI355X - the
        # chain is not host-bound (B=1, 672x672: 1.04 ms eager, 36 launches x ~10 us of host time; 1.10 ms replayed)
        self.graph_inference = os.environ.get('SSP_GRAPH_INFERENCE', '0') == '1'
        # share of the device's HBM the per-shape plan cache may hold (SSP_PLAN_MEM_FRAC).  A rebuilt plan's first step costs
        # 4-25 ms more than a cached one's (its timed choices are remembered process-wide), i.e. ~3 % of a 10-batch visit of
        # dataset.py:66-90's schedule, so a miss is cheap and the cache stays small: 0.2 peaks the 16-shape batch-64 soak at
        # 108 GB reserved (0.5, round 3: 148-157 GB) at the same images/s (profiles/r04_soak_multiscale*.json) - the peak is
        # one 832 x 832 plan (~70 GB) next to the previous shape's, which lives until the caller drops its last loss tensor
        self._plan_mem_frac = float(os.environ.get('SSP_PLAN_MEM_FRAC', '0.2'))

    # ---- network construction: same module tree as darknet.py:135-249 ----
    def create_network(self, blocks):
        models = nn.ModuleList()
        prev_filters = 3
        out_filters = []
        conv_id = 0
        for block in blocks:
            t = block['type']
            if t == 'net':
                prev_filters = int(block['channels'])
                continue
            elif t == 'convolutional':
                conv_id += 1
                bn = int(block['batch_normalize'])
                filters, k, stride = int(block['filters']), int(block['size']), int(block['stride'])
                pad = (k - 1) // 2 if int(block['pad']) else 0
                model = nn.Sequential()
                if bn:
                    model.add_module('conv{0}'.format(conv_id), nn.Conv2d(prev_filters, filters, k, stride, pad, bias=False))
                    model.add_module('bn{0}'.format(conv_id), nn.BatchNorm2d(filters, eps=1e-4))
                else:
                    model.add_module('conv{0}'.format(conv_id), nn.Conv2d(prev_filters, filters, k, stride, pad))
                # Filters are kept channels-last in memory ([Cout][kh][kw][Cin]; shape, values and the .weights format
                # are unchanged): that is the forward GEMM operand and the layout the filter gradient is accumulated in,
                # so the kernels use parameter and gradient in place instead of repacking 202 MB three times a step.
                # .cuda() / load_weights / load_state_dict / optimizers preserve it.
                # (Layers whose input channels are not a multiple of 4 - the 3-channel first layer - are padded on the
                # way to the kernels, so they keep the plain layout and the small repack.)
                conv = model[0]
                if prev_filters % 4 == 0:
                    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
                if block['activation'] == 'leaky':
                    model.add_module('leaky{0}'.format(conv_id), nn.LeakyReLU(0.1, inplace=True))
                elif block['activation'] == 'relu':
                    model.add_module('relu{0}'.format(conv_id), nn.ReLU(inplace=True))
                prev_filters = filters
                out_filters.append(prev_filters)
                models.append(model)
            elif t == 'maxpool':
                stride = int(block['stride'])
                models.append(nn.MaxPool2d(int(block['size']), stride) if stride > 1 else MaxPoolStride1())
                out_filters.append(prev_filters)
            elif t == 'avgpool':
                models.append(GlobalAvgPool2d())
                out_filters.append(prev_filters)
            elif t == 'softmax':
                models.append(nn.Softmax(dim=1))
                out_filters.append(prev_filters)
            elif t == 'cost':
                models.append(EmptyModule())
                out_filters.append(1)
            elif t == 'reorg':
                stride = int(block['stride'])
                prev_filters = stride * stride * prev_filters
                out_filters.append(prev_filters)
                models.append(Reorg(stride))
            elif t == 'route':
                ind = len(models)
                layers = resolve_layers(block['layers'], ind)
                if len(layers) == 1:
                    prev_filters = out_filters[layers[0]]
                elif len(layers) == 2:
                    assert layers[0] == ind - 1
                    prev_filters = out_filters[layers[0]] + out_filters[layers[1]]
                out_filters.append(prev_filters)
                models.append(EmptyModule())
            elif t == 'shortcut':
                ind = len(models)
                prev_filters = out_filters[ind - 1]
                out_filters.append(prev_filters)
                models.append(EmptyModule())
            elif t == 'connected':
                filters = int(block['output'])
                models.append(nn.Linear(prev_filters, filters))
                prev_filters = filters
                out_filters.append(prev_filters)
            elif t == 'region':
                loss = self._loss_cls()
                anchors = block['anchors'].split(',')
                loss.anchors = [] if anchors == [''] else [float(i) for i in anchors]
                loss.num_classes = int(block['classes'])
                loss.num_anchors = int(block['num'])
                loss.anchor_step = len(loss.anchors) // loss.num_anchors
                loss.object_scale = float(block['object_scale'])
                loss.noobject_scale = float(block['noobject_scale'])
                loss.class_scale = float(block['class_scale'])
                loss.coord_scale = float(block['coord_scale'])
                out_filters.append(prev_filters)
                models.append(loss)
            else:
                print('unknown type %s' % t)
        return models

    # ---- forward: one autograd node, HIP launches only ----
    def _plan(self, shape, device):
        """Execution plan (buffers + per-shape kernel choices) for this input shape.  Plans are kept per shape - the
        reference's multi-scale training (dataset.py:66-90) cycles through ~20 resolutions - least recently used
        first out, bounded by `_max_plans` and by `_plan_mem_frac` of the device's HBM (288 GB on MI355X)."""
        key = (shape[0], shape[1], shape[2], device.index)
        plan = self._plans.get(key)
        if plan is None:
            while len(self._plans) >= self._max_plans:
                self._plans.popitem(last=False)
            # make room BEFORE the new plan allocates (its buffers next to an evictee's were the peak of the multi-scale soak):
            # bytes per input pixel of the plans seen so far (first plan of a process: 1700, batch 64 at 416 x 416 holds 18 GB)
            budget = self._plan_mem_frac * torch.cuda.get_device_properties(device).total_memory
            px = float(shape[0] * shape[1] * shape[2])
            per_px = max([1700.0] + [p.nbytes_now() / float(p.B * p.H * p.W) for p in self._plans.values()])
            evicted = False
            while self._plans and per_px * px + sum(max(p.nbytes_est, p.nbytes_now()) for p in self._plans.values()) > budget:
                self._plans.popitem(last=False)
                evicted = True
            if evicted:
                gc.collect()                  # (a plan holds closures over itself: the cycle collector frees its tensors)
                torch.cuda.empty_cache()
            while True:
                try:
                    plan = Plan(self, shape[0], shape[1], shape[2], device)
                    break
                except torch.OutOfMemoryError:
                    if not self._plans:
                        raise
                    self._plans.popitem(last=False)
                    gc.collect()
                    torch.cuda.empty_cache()
            # forward buffers now, gradient buffers of about the same size on the first backward
            # (counted from the plan's own tensors: an allocator delta is wrong whenever the garbage collector frees another
            # model's buffers while the plan is being built)
            # (x2.5: gradient buffers of the same size, plus the Winograd operands / per-layer workspaces the tuner may add)
            plan.nbytes_est = int(2.5 * plan.footprint())
            budget = self._plan_mem_frac * torch.cuda.get_device_properties(device).total_memory
            evicted = False
            # cached plans are charged what they hold NOW (their backward and Winograd buffers exist by then)
            while self._plans and plan.nbytes_est + sum(max(p.nbytes_est, p.nbytes_now()) for p in self._plans.values()) > budget:
                self._plans.popitem(last=False)
                evicted = True
            if evicted:
                gc.collect()
                torch.cuda.empty_cache()      # hand the evicted plans' blocks back: the caching allocator would keep them
                                              # reserved next to the new plan's (a 20-shape batch-64 schedule reached 245 GB)
            red = getattr(self, '_reducer', None)
            plan.reducer = red if (red is not None and red.active) else None
            self._plans[key] = plan
        else:
            self._plans.move_to_end(key)
        return plan

    def _params(self):
        ps = []
        for m in self.models:
            if isinstance(m, nn.Sequential):
                for p in m.parameters():
                    ps.append(p)
        return ps

    def forward(self, x):
        self.loss = None   # as darknet.py:84
        if not x.is_cuda:
            raise RuntimeError("singleshotpose_amd.Darknet runs on the MI355X HIP kernels only: got a %s tensor "
                               "(no CPU fallback exists; the reference's PyTorch-CPU path lives under oracle/ as a "
                               "test checker)" % x.device)
        _lib.load()
        nc = int(self.blocks[0].get('channels', 3))
        if x.dtype == torch.uint8:
            # image bytes as the decoder yields them, (B,H,W,C): ToTensor's /255 and the NHWC layout happen on the GPU
            if x.dim() != 4 or x.size(3) != nc:
                raise ValueError("expected a (B,H,W,%d) uint8 input" % nc)
            x = x.detach().contiguous()
            shape = (x.size(0), x.size(1), x.size(2))
        else:
            if x.dim() != 4 or x.size(1) != nc:
                raise ValueError("expected a (B,%d,H,W) input" % nc)
            x = x.detach().to(torch.float32).contiguous()
            shape = (x.size(0), x.size(2), x.size(3))
        params = self._params()
        for p in params:
            if p.device != x.device:
                raise RuntimeError("model parameters are on %s but the input is on %s - call model.cuda()" % (p.device, x.device))
        plan = self._plan(shape, x.device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if need_grad and self.training:
            return _DarknetFn.apply(plan, True, x, *params)
        if need_grad:
            # eval mode with autograd on (the reference's valid.py / test() never enter no_grad: their
            # `Variable(data, volatile=True)` is a no-op today): inference-speed forward, backward by recomputation
            return _DarknetEvalFn.apply(plan, x, *params)
        if self.graph_inference and not self.training:
            return plan.forward_graph(x)      # eval: the whole launch chain as one hipGraph replay
        return plan.forward(x, self.training)

    def print_network(self):
        print_cfg(self.blocks)

    # ---- .weights I/O (darknet.py:251-394) ----
    def _read_weights(self, weightfile):
        with open(weightfile, 'rb') as fp:
            header = np.fromfile(fp, count=4, dtype=np.int32)
            buf = np.fromfile(fp, dtype=np.float32)
        self.header = torch.from_numpy(header)
        self.seen = self.header[3]
        return buf

    def _load_blocks(self, buf, nblocks):
        start = 0
        ind = -2
        for block in self.blocks[:nblocks]:
            if start >= buf.size:
                break
            ind += 1
            t = block['type']
            if t == 'convolutional':
                model = self.models[ind]
                if int(block['batch_normalize']):
                    start = load_conv_bn(buf, start, model[0], model[1])
                else:
                    start = load_conv(buf, start, model[0])
            elif t == 'connected':
                start = load_fc(buf, start, self.models[ind])
        # .data.copy_ (cfg.py's loaders, as the reference's) does not bump the autograd version counters the plans key
        # their cached filter packs / inference-mode BN constants on: invalidate them explicitly.  (Code that mutates
        # parameters through `.data` itself should call singleshotpose_amd.engine.weights_changed() too.)
        weights_changed()
        for plan in self._plans.values():
            plan.wversion.clear()
            plan.bnversion.clear()
            plan.head_budget_stale()      # the rounding budget of the forward plans was measured on the OLD weights

    def load_weights(self, weightfile):
        self._load_blocks(self._read_weights(weightfile), len(self.blocks))

    def load_weights_until_last(self, weightfile):
        # skips the last conv + region (darknet.py:310: range(blocklen-2))
        self._load_blocks(self._read_weights(weightfile), len(self.blocks) - 2)

    def save_weights(self, outfile, cutoff=0):
        if cutoff <= 0:
            cutoff = len(self.blocks) - 1
        with open(outfile, 'wb') as fp:
            self.header[3] = int(self.seen)
            self.header.numpy().tofile(fp)
            ind = -1
            for block_id in range(1, cutoff + 1):
                ind += 1
                block = self.blocks[block_id]
                if block['type'] == 'convolutional':
                    model = self.models[ind]
                    if int(block['batch_normalize']):
                        save_conv_bn(fp, model[0], model[1])
                    else:
                        save_conv(fp, model[0])
                elif block['type'] == 'connected':
                    save_fc(fp, self.models[ind])


class DarknetMulti(Darknet):
    """darknet_multi.py: same network code, region layer = the multi-object loss."""
    _loss_cls = RegionLossMulti
