"""PyTorch-CPU restatement of the reference network - TEST INFRASTRUCTURE ONLY (never imported by the product).

Follows /root/reference/darknet.py:82-130 (layer interpreter), :145-167 (conv + BatchNorm2d(eps=1e-4) +
LeakyReLU(0.1)), :168-176 (max-pool), :16-35 (Reorg), :96-106 (route), using torch.nn.functional on the CPU -
the same ATen CPU kernels the reference's own modules execute.  Pinned by tests/golden/darknet_*.npz, produced by
oracle/gen_golden.py from the reference's Darknet class itself.
"""
import numpy as np
import torch
import torch.nn.functional as F


def seeded_state(blocks, seed=0):
    """Deterministic parameters for every conv block: list (per layer) of dicts of float32 CPU tensors.

    conv weight ~ N(0, 0.02-ish scaled by fan-in), BN weight ~ U(0.5,1.5), BN bias ~ N(0,0.1),
    running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5)  (SURVEY.md section 8d, config 1).
    numpy RandomState streams are stable across numpy versions.
    """
    rs = np.random.RandomState(seed)
    state = []
    cin = int(blocks[0].get('channels', 3))
    outc = []
    for ind, b in enumerate(blocks[1:]):
        t = b['type']
        entry = None
        if t == 'convolutional':
            cout, k = int(b['filters']), int(b['size'])
            fan_in = cin * k * k
            entry = {'weight': torch.from_numpy((rs.standard_normal((cout, cin, k, k)) * (1.5 / np.sqrt(fan_in))).astype(np.float32))}
            if int(b['batch_normalize']):
                entry['bn_weight'] = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
                entry['bn_bias'] = torch.from_numpy((rs.standard_normal(cout) * 0.1).astype(np.float32))
                entry['running_mean'] = torch.from_numpy((rs.standard_normal(cout) * 0.1).astype(np.float32))
                entry['running_var'] = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
            else:
                entry['bias'] = torch.from_numpy((rs.standard_normal(cout) * 0.1).astype(np.float32))
            cin = cout
        elif t == 'reorg':
            s = int(b['stride'])
            cin = cin * s * s
        elif t == 'route':
            layers = [int(i) if int(i) > 0 else int(i) + ind for i in b['layers'].split(',')]
            cin = sum(outc[l] for l in layers)
        outc.append(cin)
        state.append(entry)
    return state


def write_weights(path, blocks, state, seen=0):
    """Darknet .weights stream (cfg.py:153-176): header int32[4], then per conv block its tensors."""
    with open(path, 'wb') as fp:
        np.array([0, 0, 0, seen], dtype=np.int32).tofile(fp)
        for b, e in zip(blocks[1:], state):
            if e is None:
                continue
            if 'bn_weight' in e:
                for k in ('bn_bias', 'bn_weight', 'running_mean', 'running_var', 'weight'):
                    e[k].numpy().tofile(fp)
            else:
                e['bias'].numpy().tofile(fp)
                e['weight'].numpy().tofile(fp)


def reorg_ref(x, stride=2):
    """darknet.py:20-35: out[b,(dy*s+dx)*C+c,hy,wx] = in[b,c,s*hy+dy,s*wx+dx]."""
    B, C, H, W = x.shape
    s = stride
    x = x.view(B, C, H // s, s, W // s, s)          # b c hy dy wx dx
    x = x.permute(0, 3, 5, 1, 2, 4).contiguous()    # b dy dx c hy wx
    return x.view(B, s * s * C, H // s, W // s)


def forward_ref(blocks, state, x, training, momentum=0.1, keep=False, raw_override=None, raws=None, tape=None,
                act_override=None, pool_override=None):
    """Runs the layer list on CPU tensors.  `state` entries may require grad; running stats are updated in place
    when training.  Returns the raw head (and every layer output when keep=True).

    raw_override {layer index: (B,Cout,H,W) tensor}: decision-frozen mode.  The VALUE of that layer's convolution output
    is replaced by the given tensor (the product's own raw conv output) while its gradient still flows into this
    layer's convolution (straight-through: override + (conv - conv.detach()), the bracket is exactly 0).  Everything
    downstream - batch statistics, leaky sign, max-pool arg-max - is then decided on identical numbers on both sides, so
    whole-network gradients can be compared at a strict tolerance instead of the fp32-vs-fp64 envelope that independent
    forward passes need (rounding flips max-pool / leaky decisions and the flips propagate).
    raws (dict, optional): filled with this function's own convolution outputs {layer index: tensor (detached)}.
    tape (dict, optional): filled with {layer index: (conv input, conv output node with retain_grad, padding)} so a
    caller can re-evaluate a filter gradient in float64 after backward (step_check.py: ill-conditioned sums).
    act_override {layer index: (B,Cout,H,W) tensor}: the product's leaky(BN(conv)) output of an un-pooled block.  Only its
    SIGN is used: the leaky branch of every element is the one the product took (y > 0 there), while value and gradient
    stay this function's own.  With frozen raw outputs the pre-activation y = BN(raw) still differs between two fp32
    evaluations in its last bits (scale * raw + shift against (raw - mean) * invstd * gamma + beta), so an element
    within ~1e-7 of zero can take different branches; when that element carries a large share of the gradient (the
    label's cell in the deep layers) one flipped branch moves a channel's gradient by 1e-3.  Both branches are valid
    fp32 results - freezing the decision removes the ambiguity from the comparison.
    pool_override {max-pool layer index: int64 (B,C,Ho,Wo) flat indices as F.max_pool2d(return_indices=True) yields}:
    the window element the product's max-pool selected (first maximum of ITS activations); this function's activation
    is gathered there instead of re-deciding the maximum on its own last bits."""
    outputs = {}
    for ind, b in enumerate(blocks[1:]):
        t = b['type']
        if t == 'convolutional':
            e = state[ind]
            k = int(b['size'])
            pad = (k - 1) // 2 if int(b['pad']) else 0
            x_in = x
            x = F.conv2d(x, e['weight'], e.get('bias'), stride=int(b['stride']), padding=pad)
            if raws is not None:
                raws[ind] = x.detach()
            if raw_override is not None and ind in raw_override:
                x = raw_override[ind] + (x - x.detach())
            if tape is not None and x.requires_grad:
                x.retain_grad()
                tape[ind] = (x_in, x, pad)
            if 'bn_weight' in e:
                x = F.batch_norm(x, e['running_mean'], e['running_var'], e['bn_weight'], e['bn_bias'], training,
                                 momentum, 1e-4)
            if b['activation'] == 'leaky':
                if act_override is not None and ind in act_override:
                    x = torch.where(act_override[ind] > 0, x, x * 0.1)
                else:
                    x = F.leaky_relu(x, 0.1)
            elif b['activation'] == 'relu':
                x = F.relu(x)
        elif t == 'maxpool':
            if pool_override is not None and ind in pool_override:
                idx = pool_override[ind]
                x = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
            else:
                x = F.max_pool2d(x, int(b['size']), int(b['stride']))
        elif t == 'reorg':
            x = reorg_ref(x, int(b['stride']))
        elif t == 'route':
            layers = [int(i) if int(i) > 0 else int(i) + ind for i in b['layers'].split(',')]
            x = outputs[layers[0]] if len(layers) == 1 else torch.cat([outputs[l] for l in layers], 1)
        elif t in ('region', 'cost'):
            continue
        else:
            raise NotImplementedError(t)
        outputs[ind] = x
    return (x, outputs) if keep else x
