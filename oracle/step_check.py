"""Whole-training-step checker: the product's HIP step against the CPU oracle - TEST INFRASTRUCTURE ONLY.

Used by tests/test_gpu_fullsize.py, __graft_entry__.smoke() and bench.py's --verify leg (outside every timed region).
One call runs ONE training step of a product Darknet on the GPU (forward, RegionLoss, backward - what
/root/reference/train.py:83-103 does per batch) and the same step on the oracle (oracle/darknet_ref.py,
oracle/region_loss_ref.py: the reference's PyTorch-CPU semantics), from the same weights, batch and labels, and
returns the error of every quantity the step produces:

  head      max|a-b|/max|b| of the raw network output vs an independent oracle forward     (bar 1e-4)
  head64 / head64_ref (exact=True)  the product's head and the fp32 oracle's head against an independent FLOAT64 forward
  loss      relative error of the RegionLoss value                                          (bar 1e-4)
  running   worst max-normalised error of the BatchNorm running_mean / running_var updates  (bar 1e-4)
  conv      worst per-layer error of the product's raw conv outputs vs the oracle's convolution of the product's
            own layer inputs (layer-local, every conv launch of the step with the plan the autotuner picked)
  grad      worst per-parameter max-normalised error of the parameter gradients against the DECISION-FROZEN oracle
            backward (forward_ref(raw_override=..., act_override=...): the oracle's autograd runs on the product's own
            raw conv outputs, so batch statistics and max-pool winners are decided on identical numbers, and every
            element of an un-pooled leaky block takes the branch the product took; strict bar).  A filter
            gradient whose fp32 oracle value is itself inexact (ill-conditioned sum) is compared against the float64
            re-evaluation of the oracle's own operands instead ('grad_fp64_oracle' lists those and all three errors)
  grad_out  error of dL/d(head) (the RegionLoss gradient) against the oracle's on the product's head
  grad64 / grad64_ref (exact=True)  the same raw-output-frozen network once more in FLOAT64 (its BatchNorm arithmetic,
            and therefore the leaky signs / pool winners of elements within fp32 rounding of a decision boundary, are
            float64's own) and against it the worst per-parameter distance of the product ('grad64') and of the fp32
            oracle itself ('grad64_ref').  A yardstick, not a tighter oracle: it shows how far fp32 arithmetic of ANY
            summation order sits from float64 on this network (1e-3 .. 1e-2 on some parameters) and that the product
            sits exactly where the reference does, parameter by parameter ('grad64_by_param' holds the pairs)

SURVEY.md section 8(d) config 2 / config 5; tolerance north_star: fp32 conv / loss within 1e-4 relative.
"""
import numpy as np
import torch

from .darknet_ref import forward_ref
from .region_loss_ref import region_loss_ref


def _rel(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def snapshot_state(model):
    """Product module tree -> oracle state list (CPU clones), in block order."""
    state = []
    for ind, block in enumerate(model.blocks[1:]):
        if block['type'] != 'convolutional':
            state.append(None)
            continue
        seq = model.models[ind]
        e = {'weight': seq[0].weight.detach().cpu().contiguous().clone()}
        if int(block['batch_normalize']):
            bn = seq[1]
            e['bn_weight'] = bn.weight.detach().cpu().clone()
            e['bn_bias'] = bn.bias.detach().cpu().clone()
            e['running_mean'] = bn.running_mean.detach().cpu().clone()
            e['running_var'] = bn.running_var.detach().cpu().clone()
        else:
            e['bias'] = seq[0].bias.detach().cpu().clone()
        state.append(e)
    return state


def _clone(state, requires_grad=False):
    out = []
    for e in state:
        if e is None:
            out.append(None)
            continue
        d = {}
        for k, v in e.items():
            t = v.clone()
            if requires_grad and not k.startswith('running'):
                t.requires_grad_(True)
            d[k] = t
        out.append(d)
    return out


def check_train_step(model, crit, x_cpu, tgt, epoch, loss_kwargs=None, frozen_backward=True, exact=False,
                     first_raw_slice_bytes=(1 << 31) - 1):
    """model: product Darknet on the GPU (train mode is set here); crit: product RegionLoss(Multi); x_cpu (B,3,H,W)
    float32 CPU; tgt (B, 50*21) CPU labels.  Returns a dict of errors (see the module docstring) plus 'plans', the
    (layer, forward plan code, dgrad plan code) triples that were active."""
    loss_kwargs = dict(loss_kwargs or {})
    dev = next(model.parameters()).device
    state0 = snapshot_state(model)
    model.train()
    model.zero_grad(set_to_none=True)
    verbose, crit.verbose = crit.verbose, False
    out = model(x_cpu.to(dev))
    B, H, W = x_cpu.size(0), x_cpu.size(2), x_cpu.size(3)
    plan = model._plans[(B, H, W, dev.index)]
    # the product's raw conv outputs, NHWC [M][coutp] -> NCHW CPU (backward rewrites them in place: fetch now)
    raws = {}
    first_raw = None
    for ind, cs in plan.convs.items():
        if getattr(cs, 'first_live', False):
            # first block, fused form: its raw conv output is recomputed by every pass, never stored.  ssp_first_conv_raw
            # evaluates it once more with the same instruction sequence (bit-identical values) for this checker.
            from singleshotpose_amd import _lib
            tmp = torch.empty(cs.M * cs.cout, dtype=torch.float32, device=dev)
            # (in batch slices: the entry point addresses its output with 32-bit offsets - 2 GiB per call; batch 64 at 608 x 608 is 3 GB)
            per = max(1, min(B, first_raw_slice_bytes // (cs.H * cs.W * cs.cout * 4)))
            for b0 in range(0, B, per):
                nb = min(per, B - b0)
                _lib.call('ssp_first_conv_raw', cs.inp.ptr + 4 * b0 * cs.H * cs.W * cs.inp.ld, plan._wbuf(cs).data_ptr(),
                          tmp.data_ptr() + 4 * b0 * cs.H * cs.W * cs.cout, cs.cout, nb, cs.H, cs.W,
                          torch.cuda.current_stream().cuda_stream)
            raws[ind] = tmp.view(B, cs.H, cs.W, cs.cout).permute(0, 3, 1, 2).contiguous().cpu()
            first_raw = (ind, tmp)      # kept on the device: its decisions are frozen below like every other pooled block's
            continue
        r = cs.raw.view(B, cs.H, cs.W, cs.ldraw)[..., :cs.cout].permute(0, 3, 1, 2)
        raws[ind] = r.contiguous().cpu()
    # the product's activations of the un-pooled leaky blocks: their SIGN freezes the leaky branch per element in the
    # oracle's backward (forward_ref(act_override=...)); the pooled blocks keep the oracle's own decisions
    acts = {}
    for ind, cs in plan.convs.items():
        if cs.needs_act and not cs.pool and cs.slope == 0.1 and not getattr(cs, 'first_live', False):
            a = cs.out
            acts[ind] = a.t[a.off:].view(-1)[:B * a.H * a.W * a.ld].view(B, a.H, a.W, a.ld)[..., :cs.cout] \
                .permute(0, 3, 1, 2).contiguous().cpu() if a.off == 0 else None
    acts = {k: v for k, v in acts.items() if v is not None}
    # pooled blocks (the fused first one, whose raw output is never stored, follows below): the product's own BN + leaky kernel once
    # more WITHOUT the pooling, into a scratch buffer - its signs freeze the leaky branches, its first-maximum positions the
    # pool winners (forward_ref(pool_override=...)).  Same kernel, same scale / shift vectors, same expression as the
    # pooled launch of the forward pass and the recomputation in backward.
    pools = {}
    import torch.nn.functional as F
    from singleshotpose_amd import _lib
    for ind, cs in plan.convs.items():
        if cs.pool and cs.needs_act and cs.slope == 0.1 and not getattr(cs, 'first_live', False) and cs.coutp == cs.cout:
            scratch = torch.empty(cs.M * cs.coutp, dtype=torch.float32, device=dev)
            v = cs.vec
            _lib.call('ssp_bn_act_fwd', cs.raw.data_ptr(), cs.ldraw, scratch.data_ptr(), cs.coutp, v[2].data_ptr(),
                      v[3].data_ptr(), cs.coutp, B, cs.H, cs.W, 0, cs.slope, torch.cuda.current_stream().cuda_stream)
            a = scratch.view(B, cs.H, cs.W, cs.coutp).permute(0, 3, 1, 2).contiguous().cpu()
            acts[ind] = a
            pools[ind + 1] = F.max_pool2d(a, 2, 2, return_indices=True)[1]
            del scratch
    if first_raw is not None:
        # ... and the fused first block: the same BN + leaky expression (scale * raw + shift, leaky) over the raw values its
        # four passes recompute (ssp_first_conv_raw: bit-identical), un-pooled, in 16-image slices (the map is 1.4 GB at batch
        # 64): leaky signs and first-maximum pool winners of the product.  Without this the oracle took its OWN decisions on
        # the first block, and a handful of flipped elements among its tens of millions moved the first block's
        # BatchNorm-bias gradient by 1e-4 ... 1e-2 at some shapes (tools/multiscale_check.py, 288 and 544 at batch 8).
        ind, tmp = first_raw
        cs = plan.convs[ind]
        if cs.pool and cs.needs_act and cs.slope == 0.1:
            v = cs.vec
            parts = []
            step_b = 16
            for b0 in range(0, B, step_b):
                nb = min(step_b, B - b0)
                scratch = torch.empty(nb * cs.H * cs.W * cs.cout, dtype=torch.float32, device=dev)
                _lib.call('ssp_bn_act_fwd', tmp.data_ptr() + 4 * b0 * cs.H * cs.W * cs.cout, cs.cout, scratch.data_ptr(), cs.cout,
                          v[2].data_ptr(), v[3].data_ptr(), cs.cout, nb, cs.H, cs.W, 0, cs.slope,
                          torch.cuda.current_stream().cuda_stream)
                parts.append(scratch.view(nb, cs.H, cs.W, cs.cout).permute(0, 3, 1, 2).contiguous().cpu())
                del scratch
            a = torch.cat(parts, 0)
            acts[ind] = a
            pools[ind + 1] = F.max_pool2d(a, 2, 2, return_indices=True)[1]
        del tmp
    loss = crit(out, tgt, epoch)
    loss.backward()
    torch.cuda.synchronize()
    crit.verbose = verbose
    out_c = out.detach().cpu()
    res = {'plans': [(ind, cs.plan_fwd, cs.plan_dgrad) for ind, cs in sorted(plan.convs.items())]}

    # ---- independent oracle forward: head, loss, running statistics ----
    st_a = _clone(state0)
    with torch.no_grad():
        y_ref = forward_ref(model.blocks, st_a, x_cpu, training=True)
    r_ref = region_loss_ref(y_ref, tgt, epoch, **loss_kwargs)
    res['head'] = _rel(out_c, y_ref)
    res['loss'] = abs(float(loss) - r_ref['loss']) / max(abs(r_ref['loss']), 1e-30)
    res['loss_gpu'], res['loss_ref'] = float(loss), r_ref['loss']
    run = 0.0
    for ind, e in enumerate(st_a):
        if e is not None and 'running_mean' in e:
            bn = model.models[ind][1]
            run = max(run, _rel(bn.running_mean.cpu(), e['running_mean']), _rel(bn.running_var.cpu(), e['running_var']))
    res['running'] = run
    if exact:
        # float64 yardstick of the FORWARD pass: an independent float64 evaluation of the same network on the same batch, and
        # against it the product's head ('head64') and the fp32 oracle's own ('head64_ref': PyTorch-CPU float32 sits ~2e-5 of
        # the head's range from float64 on this network - BatchNorm amplifies every block's rounding on the way down,
        # tools/head_amplification.py - so 'head', the distance between two fp32 evaluations, cannot go below that)
        st64 = [None if e is None else {k: v.double() for k, v in e.items()} for e in state0]
        with torch.no_grad():
            y64 = forward_ref(model.blocks, st64, x_cpu.double(), training=True)
        res['head64'], res['head64_ref'] = _rel(out_c, y64), _rel(y_ref, y64)
        del st64, y64
    if not frozen_backward:
        return res

    # ---- decision-frozen oracle: per-layer conv check + whole-network gradients ----
    st_b = _clone(state0, requires_grad=True)
    own, tape = {}, {}
    y_frozen = forward_ref(model.blocks, st_b, x_cpu, training=True, raw_override=raws, raws=own, tape=tape,
                           act_override=acts, pool_override=pools)
    res['frozen_leaky_layers'], res['frozen_pool_layers'] = len(acts), len(pools)
    res['conv_by_layer'] = {ind: _rel(raws[ind], own[ind]) for ind in sorted(raws)}
    res['conv'] = max(res['conv_by_layer'].values())
    r_frz = region_loss_ref(out_c, tgt, epoch, **loss_kwargs)        # loss gradient on the product's own head
    y_frozen.backward(r_frz['grad'])
    # dL/d(head) of the product = gradient of the first backward node; recompute it through the product's loss
    o2 = out.detach().clone().requires_grad_(True)
    crit.verbose = False
    crit(o2, tgt, epoch).backward()
    crit.verbose = verbose
    res['grad_out'] = _rel(o2.grad.cpu(), r_frz['grad'])
    gerr = {}
    for ind, e in enumerate(st_b):
        if e is None:
            continue
        seq = model.models[ind]
        mine = seq[0].weight.grad.cpu()
        err = _rel(mine, e['weight'].grad)
        first_conv = ind == min(i for i, e_ in enumerate(st_b) if e_ is not None)
        if (err > 1e-4 or first_conv) and ind in tape:
            # An ill-conditioned filter gradient (the first layer's is sum dx * image with sum dx = 0 exactly and an
            # all-positive image: the terms cancel ~1e3 : 1 at B = 64) - the oracle's OWN fp32 accumulation (oneDNN) sits
            # >1e-4 from the exact sum of its own operands.  Re-evaluate the oracle's gradient from the very same
            # (input, dL/d raw) tensors in float64 and compare against that: a tighter oracle, not a looser bar.  The first
            # layer's filter gradient is ALWAYS judged this way (round 5): against the fp32 oracle its "error" read 3e-5 ...
            # 1e-4 from shape to shape - the oracle's own rounding; against the float64 sum the product sits at ~1e-5.
            x_in, node, pad = tape[ind]
            g64 = torch.nn.grad.conv2d_weight(x_in.detach().double(), e['weight'].shape, node.grad.double(), padding=pad)
            res.setdefault('grad_fp64_oracle', {})['%d.weight' % ind] = dict(
                vs_fp32_oracle=err, oracle_fp32_vs_fp64=_rel(e['weight'].grad, g64), vs_fp64_oracle=_rel(mine, g64))
            err = _rel(mine, g64)
        gerr['%d.weight' % ind] = err
        if 'bn_weight' in e:
            gerr['%d.bn_weight' % ind] = _rel(seq[1].weight.grad.cpu(), e['bn_weight'].grad)
            gerr['%d.bn_bias' % ind] = _rel(seq[1].bias.grad.cpu(), e['bn_bias'].grad)
        else:
            gerr['%d.bias' % ind] = _rel(seq[0].bias.grad.cpu(), e['bias'].grad)
    res['grad_by_param'] = gerr
    res['grad'] = max(gerr.values())
    if exact:
        res.update(exact_frozen_errors(model, state0, st_b, x_cpu, raws, r_frz['grad'], acts, pools))
    return res


def exact_frozen_errors(model, state0, st32, x_cpu, raws, grad_head, acts=None, pools=None):
    """Float64 evaluation of the raw-output-frozen network (same overrides: the product's raw conv outputs, exactly
    representable in float64; batch statistics - and with them the leaky sign / pool winner of the few elements that sit
    within fp32 rounding of a decision boundary - are float64's) and, against it, the per-parameter distances of the
    product's gradients and of the fp32 oracle's (st32 after its backward)."""
    st64 = []
    for e in state0:
        if e is None:
            st64.append(None)
            continue
        d = {}
        for k, v in e.items():
            d[k] = v.double().clone()
            if not k.startswith('running'):
                d[k].requires_grad_(True)
        st64.append(d)
    y64 = forward_ref(model.blocks, st64, x_cpu.double(), training=True,
                      raw_override={k: v.double() for k, v in raws.items()}, act_override=acts, pool_override=pools)
    y64.backward(grad_head.double())
    pairs = {}
    for ind, e in enumerate(st64):
        if e is None:
            continue
        seq = model.models[ind]
        names = [('weight', seq[0].weight)]
        if 'bn_weight' in e:
            names += [('bn_weight', seq[1].weight), ('bn_bias', seq[1].bias)]
        else:
            names += [('bias', seq[0].bias)]
        for k, prm in names:
            g64 = e[k].grad
            pairs['%d.%s' % (ind, k)] = (_rel(prm.grad.cpu(), g64), _rel(st32[ind][k].grad, g64))
    return {'grad64_by_param': pairs, 'grad64': max(a for a, _ in pairs.values()),
            'grad64_ref': max(b for _, b in pairs.values())}


def summarize(res):
    keys = [k for k in ('head', 'head64', 'head64_ref', 'loss', 'running', 'conv', 'grad_out', 'grad', 'grad64', 'grad64_ref') if k in res]
    return ', '.join('%s %.2e' % (k, res[k]) for k in keys)
