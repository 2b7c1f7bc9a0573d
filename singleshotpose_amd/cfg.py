"""Darknet .cfg parser and .weights (de)serialisers - host-side mirror of the reference's cfg.py.

Same public names and argument meaning as /root/reference/cfg.py:
  parse_cfg (cfg.py:4-34), print_cfg (cfg.py:36-151), load_conv / save_conv / load_conv_bn / save_conv_bn /
  load_fc / save_fc (cfg.py:153-201).

File formats:
  .cfg      "[section]" starts a block, "key=value" lines (both sides stripped), '#' comments; every value stays a
            string; a key literally called "type" is stored as "_type"; convolutional blocks default
            batch_normalize to 0.
  .weights  float32 stream; per conv block with BN: bn.bias, bn.weight, running_mean, running_var, conv.weight;
            without BN: conv.bias, conv.weight; conv.weight flattened in (Cout, Cin, kh, kw) order.
"""
import numpy as np
import torch


def parse_cfg(cfgfile):
    """cfg text -> list of str->str dicts; block 0 is [net]."""
    blocks = []
    current = None
    with open(cfgfile, 'r') as fp:
        for raw in fp:
            line = raw.rstrip()
            if line == '' or line[0] == '#':
                continue
            if line[0] == '[':
                if current:
                    blocks.append(current)
                current = {'type': line.lstrip('[').rstrip(']')}
                if current['type'] == 'convolutional':
                    current['batch_normalize'] = 0
                continue
            key, value = line.split('=')
            key = key.strip()
            current['_type' if key == 'type' else key] = value.strip()
    if current:
        blocks.append(current)
    return blocks


def resolve_layers(spec, ind):
    """route 'layers=' entries: positive = absolute index, otherwise relative to the current layer (darknet.py:98)."""
    return [int(i) if int(i) > 0 else int(i) + ind for i in spec.split(',')]


def layer_shapes(blocks, width=None, height=None):
    """(width, height, filters) of every layer's output, index-aligned with Darknet.models (block 0 excluded).

    Follows the bookkeeping of cfg.py:36-151 / darknet.py:135-249.
    """
    net = blocks[0]
    w = int(net['width']) if width is None else width
    h = int(net['height']) if height is None else height
    c = int(net.get('channels', 3))
    out = []
    for ind, block in enumerate(blocks[1:]):
        t = block['type']
        if t == 'convolutional':
            k, s = int(block['size']), int(block['stride'])
            pad = (k - 1) // 2 if int(block['pad']) else 0
            w = (w + 2 * pad - k) // s + 1
            h = (h + 2 * pad - k) // s + 1
            c = int(block['filters'])
        elif t == 'maxpool':
            s = int(block['stride'])
            w, h = w // s, h // s
        elif t == 'avgpool':
            w, h = 1, 1
        elif t == 'reorg':
            s = int(block['stride'])
            w, h, c = w // s, h // s, c * s * s
        elif t == 'route':
            layers = resolve_layers(block['layers'], ind)
            w, h = out[layers[0]][0], out[layers[0]][1]
            c = sum(out[l][2] for l in layers)
        elif t == 'shortcut':
            f = int(block['from'])
            f = f if f > 0 else f + ind
            w, h, c = out[f]
        elif t == 'connected':
            w, h, c = 1, 1, int(block['output'])
        out.append((w, h, c))
    return out


def print_cfg(blocks):
    """Layer table in the reference's format (README.md:74-81 shows the expected text)."""
    print('layer     filters    size              input                output')
    net = blocks[0]
    shapes = layer_shapes(blocks)
    pw, ph, pc = int(net['width']), int(net['height']), 3
    for ind, block in enumerate(blocks[1:]):
        t = block['type']
        w, h, c = shapes[ind]
        if t == 'convolutional':
            k, s = int(block['size']), int(block['stride'])
            print('%5d %-6s %4d  %d x %d / %d   %3d x %3d x%4d   ->   %3d x %3d x%4d' % (ind, 'conv', c, k, k, s, pw, ph, pc, w, h, c))
        elif t == 'maxpool':
            k, s = int(block['size']), int(block['stride'])
            print('%5d %-6s       %d x %d / %d   %3d x %3d x%4d   ->   %3d x %3d x%4d' % (ind, 'max', k, k, s, pw, ph, pc, w, h, c))
        elif t == 'avgpool':
            print('%5d %-6s                   %3d x %3d x%4d   ->  %3d' % (ind, 'avg', pw, ph, pc, pc))
        elif t == 'softmax':
            print('%5d %-6s                                    ->  %3d' % (ind, 'softmax', pc))
        elif t == 'cost':
            print('%5d %-6s                                     ->  %3d' % (ind, 'cost', pc))
        elif t == 'reorg':
            print('%5d %-6s             / %d   %3d x %3d x%4d   ->   %3d x %3d x%4d' % (ind, 'reorg', int(block['stride']), pw, ph, pc, w, h, c))
        elif t == 'route':
            layers = resolve_layers(block['layers'], ind)
            print(('%5d %-6s' + ' %d' * len(layers)) % ((ind, 'route') + tuple(layers)))
        elif t == 'region':
            print('%5d %-6s' % (ind, 'detection'))
        elif t == 'shortcut':
            f = int(block['from'])
            print('%5d %-6s %d' % (ind, 'shortcut', f if f > 0 else f + ind))
        elif t == 'connected':
            print('%5d %-6s                            %d  ->  %3d' % (ind, 'connected', pc, c))
        else:
            print('unknown type %s' % t)
        pw, ph, pc = w, h, c


def _take(buf, start, tensor):
    n = tensor.numel()
    tensor.copy_(torch.from_numpy(buf[start:start + n]).view_as(tensor))
    return start + n


def _dump(fp, tensor):
    tensor.detach().to('cpu', torch.float32).contiguous().numpy().tofile(fp)


def load_conv(buf, start, conv_model):
    start = _take(buf, start, conv_model.bias.data)
    return _take(buf, start, conv_model.weight.data)


def save_conv(fp, conv_model):
    _dump(fp, conv_model.bias.data)
    _dump(fp, conv_model.weight.data)


def load_conv_bn(buf, start, conv_model, bn_model):
    start = _take(buf, start, bn_model.bias.data)
    start = _take(buf, start, bn_model.weight.data)
    start = _take(buf, start, bn_model.running_mean)
    start = _take(buf, start, bn_model.running_var)
    return _take(buf, start, conv_model.weight.data)


def save_conv_bn(fp, conv_model, bn_model):
    _dump(fp, bn_model.bias.data)
    _dump(fp, bn_model.weight.data)
    _dump(fp, bn_model.running_mean)
    _dump(fp, bn_model.running_var)
    _dump(fp, conv_model.weight.data)


def load_fc(buf, start, fc_model):
    start = _take(buf, start, fc_model.bias.data)
    return _take(buf, start, fc_model.weight.data)


def save_fc(fp, fc_model):
    _dump(fp, fc_model.bias.data)
    _dump(fp, fc_model.weight.data)


if __name__ == '__main__':
    import sys
    print_cfg(parse_cfg(sys.argv[1] if len(sys.argv) == 2 else 'cfg/yolo-pose.cfg'))
