"""TEST INFRASTRUCTURE ONLY -- independent functional restatement of the Mannequin-Challenge
hourglass on plain torch.nn.functional ops, CPU or any device.

PARITY UNPINNED for the CNN: the network source is an un-vendored submodule
(roxanneluo/mannequinchallenge, /root/reference/.gitmodules:4-6, SHA unknown) and the
reference holds no test vectors for it.  This restatement follows the published
google/mannequinchallenge models/hourglass.py as summarised in SURVEY.md A.3; it is written
separately from consistent_depth_amd/monodepth/hourglass.py (recursive functions over the
state_dict instead of nn.Module composition) so the two cross-check each other, and it is the
fp32/fp64 reference the HIP layers are compared with.

forward(state, x, training) -> (pred_d, pred_confidence); `state` is a dict with the upstream
checkpoint keys (seq.0.weight, seq.3.list.0.1.convs.1.3.weight, ..., pred_layer.bias).
In training mode BatchNorm uses batch statistics and (optionally) updates running stats in
`state` in place, exactly like nn.BatchNorm2d(momentum=0.1, eps=1e-5).
"""
import torch
import torch.nn.functional as F

from .conv64 import conv2d_same      # fp64 on the CPU: dgemm-based (torch's double conv2d has no vendor kernel); else F.conv2d

_E = [[64], [3, 32, 64], [5, 32, 64], [7, 32, 64]]
_SPEC = {
    "A": [[16], [3, 32, 16], [7, 32, 16], [11, 32, 16]],
    "A2": [[16], [3, 64, 16], [7, 64, 16], [11, 64, 16]],
    "B": [[32], [3, 32, 32], [5, 32, 32], [7, 32, 32]],
    "B2": [[32], [3, 64, 32], [5, 64, 32], [7, 64, 32]],
    "C": [[32], [3, 64, 32], [7, 64, 32], [11, 64, 32]],
    "D": _E, "E": _E,
    "F": [[64], [3, 64, 64], [7, 64, 64], [11, 64, 64]],
    "G": [[32], [3, 32, 32], [5, 32, 32], [7, 32, 32]],
}
# (left branch, right branch) of Channels{1..4}; ints recurse
_TREE = {
    1: (["E", "E"], ["P", "E", "E", "E", "U"]),
    2: (["E", "F"], ["P", "E", "E", 1, "E", "F", "U"]),
    3: (["P", "B", "D", 2, "E", "G", "U"], ["B", "C"]),
    4: (["P", "B", "B", 3, "B2", "A", "U"], ["A2"]),
}


def _bn(state, key, x, training, affine, update):
    rm, rv = state.get(key + ".running_mean"), state.get(key + ".running_var")
    w = state[key + ".weight"] if affine else None
    b = state[key + ".bias"] if affine else None
    if training and not update and rm is not None:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, w, b, training, 0.1, 1e-5)
    if training and update and (key + ".num_batches_tracked") in state:
        state[key + ".num_batches_tracked"] += 1
    return y


def _conv_bn_relu(state, key_conv, key_bn, x, pad, training, update):
    assert pad == (state[key_conv + ".weight"].shape[-1] - 1) // 2
    x = conv2d_same(x, state[key_conv + ".weight"], state[key_conv + ".bias"])
    return F.relu(_bn(state, key_bn, x, training, False, update))


def _inception(state, pre, kind, x, training, update):
    outs = [_conv_bn_relu(state, f"{pre}.convs.0.0", f"{pre}.convs.0.1", x, 0, training, update)]
    for i, (k, _mid, _out) in enumerate(_SPEC[kind][1:], start=1):
        h = _conv_bn_relu(state, f"{pre}.convs.{i}.0", f"{pre}.convs.{i}.1", x, 0, training, update)
        outs.append(_conv_bn_relu(state, f"{pre}.convs.{i}.3", f"{pre}.convs.{i}.4", h, (k - 1) // 2, training, update))
    return torch.cat(outs, 1)


def _channels(state, pre, level, x, training, update):
    total = None
    for side, items in enumerate(_TREE[level]):
        h = x
        for j, it in enumerate(items):
            p = f"{pre}.list.{side}.{j}"
            if it == "P":
                h = F.avg_pool2d(h, 2)
            elif it == "U":
                h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
            elif isinstance(it, int):
                h = _channels(state, p, it, h, training, update)
            else:
                h = _inception(state, p, it, h, training, update)
        total = h if total is None else total + h
    return total


def forward(state, x, training=True, update_running_stats=False):
    h = conv2d_same(x, state["seq.0.weight"], state["seq.0.bias"])
    h = F.relu(_bn(state, "seq.1", h, training, True, update_running_stats))
    feat = _channels(state, "seq.3", 4, h, training, update_running_stats)
    pred = conv2d_same(feat, state["pred_layer.weight"], state["pred_layer.bias"])
    conf = torch.sigmoid(conv2d_same(feat, state["uncertainty_layer.0.weight"], state["uncertainty_layer.0.bias"]))
    return pred, conf


def conv_macs(H, W):
    """Forward multiply-accumulates of all convolutions for one HxW image."""
    total = [0]

    def inc(cin, spec, h, w):
        total[0] += h * w * cin * spec[0][0]
        for k, mid, out in spec[1:]:
            total[0] += h * w * (cin * mid + mid * k * k * out)

    cin_of = {"A": 128, "A2": 128, "B": 128, "B2": 128, "C": 128, "D": 128, "E": 256, "F": 256, "G": 256}

    def walk(level, h, w):
        for items in _TREE[level]:
            hh, ww = h, w
            for it in items:
                if it == "P":
                    hh, ww = hh // 2, ww // 2
                elif it == "U":
                    hh, ww = hh * 2, ww * 2
                elif isinstance(it, int):
                    walk(it, hh, ww)
                else:
                    inc(cin_of[it], _SPEC[it], hh, ww)

    total[0] += H * W * 3 * 49 * 128
    walk(4, H, W)
    total[0] += 2 * H * W * 64 * 9
    return total[0]
