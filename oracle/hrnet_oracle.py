"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference HRNetV2 segmentation net.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates /root/reference/uncertainty_modeling/models/hrnet_module.py as a pure function of (config, state dict, x):
  * BasicBlock.forward            (:59-77)    conv3x3-BN-ReLU-conv3x3-BN (+downsample) + residual, ReLU
  * Bottleneck.forward            (:99-119)   1x1-BN-ReLU, 3x3-BN-ReLU, 1x1-BN (+downsample) + residual, ReLU
  * HighResolutionModule.forward  (:308-334)  branches, then SUM fusion: j>i 1x1+BN then bilinear up to branch i;
                                              j<i chain of stride-2 3x3+BN(+ReLU except last); ReLU
  * HighResolutionNet.forward     (:597-671)  stem, layer1, transitions, stages 2-4, optional F.dropout(0.5,
                                              training=True) on the 4 stage-4 outputs (DROPOUT_FINAL), bilinear
                                              upsample + concat, last_layer (1x1+BN+ReLU+conv), bilinear to input size
BatchNorm runs in TRAINING mode (batch statistics, biased variance, eps 1e-5): the reference never calls .eval()
(SURVEY D5), so every BN normalises with the statistics of the current batch.  The running-stat side effect never
influences outputs and is not restated.  Dropout masks are injected (captured from the reference run).

Parity pin: tests/test_oracle_golden.py vs tests/golden/hrnet_small.npz (outputs of the imported reference class).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

ALIGN_CORNERS = False  # configs/model/hrnet_config*.yaml: ALIGN_CORNERS: False


def _bn(x, sd, name):
    return F.batch_norm(x, None, None, sd[name + ".weight"], sd[name + ".bias"], training=True, momentum=0.1, eps=1e-5)


def _conv(x, sd, name, stride=1, pad=None):
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    if pad is None:
        pad = w.shape[-1] // 2
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def _basic(x, sd, p):
    out = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"))
    out = _bn(_conv(out, sd, p + ".conv2"), sd, p + ".bn2")
    res = x
    if p + ".downsample.0.weight" in sd:
        res = _bn(_conv(x, sd, p + ".downsample.0"), sd, p + ".downsample.1")
    return F.relu(out + res)


def _bottleneck(x, sd, p):
    out = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"))
    out = F.relu(_bn(_conv(out, sd, p + ".conv2"), sd, p + ".bn2"))
    out = _bn(_conv(out, sd, p + ".conv3"), sd, p + ".bn3")
    res = x
    if p + ".downsample.0.weight" in sd:
        res = _bn(_conv(x, sd, p + ".downsample.0"), sd, p + ".downsample.1")
    return F.relu(out + res)


def _block(x, sd, p, kind):
    return _bottleneck(x, sd, p) if kind == "BOTTLENECK" else _basic(x, sd, p)


def _module(xs, sd, p, nb, nblocks, kind):
    xs = list(xs)
    for i in range(nb):
        for b in range(nblocks[i]):
            xs[i] = _block(xs[i], sd, f"{p}.branches.{i}.{b}", kind)
    if nb == 1:
        return xs
    out = []
    for i in range(nb):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                t = _bn(_conv(xs[j], sd, f"{p}.fuse_layers.{i}.{j}.0"), sd, f"{p}.fuse_layers.{i}.{j}.1")
                t = F.interpolate(t, size=xs[i].shape[-2:], mode="bilinear", align_corners=ALIGN_CORNERS)
            else:
                t = xs[j]
                for k in range(i - j):
                    q = f"{p}.fuse_layers.{i}.{j}.{k}"
                    t = _bn(_conv(t, sd, q + ".0", stride=2), sd, q + ".1")
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        out.append(F.relu(y))
    return out


def _transition(prev, sd, p, n_prev, n_cur):
    out = []
    for i in range(n_cur):
        if i < n_prev:
            if f"{p}.{i}.0.weight" in sd:
                out.append(F.relu(_bn(_conv(prev[i], sd, f"{p}.{i}.0"), sd, f"{p}.{i}.1")))
            else:
                out.append(prev[i])
        else:
            t = prev[-1]
            for j in range(i + 1 - n_prev):
                t = F.relu(_bn(_conv(t, sd, f"{p}.{i}.{j}.0", stride=2), sd, f"{p}.{i}.{j}.1"))
            out.append(t)
    return out


def hrnet_forward(extra, sd, x, dropout_masks=None, return_features=False, ssn=None):
    """extra: the MODEL.EXTRA dict of the yaml (STAGE1..4, FINAL_CONV_KERNEL, optional DROPOUT_FINAL);
    sd: state dict (reference key names) of torch tensors; x (B,Cin,H,W).
    dropout_masks: None or 4 bool keep-masks (stage-4 outputs) for DROPOUT_FINAL.
    ssn: None, or (num_classes, rank, epsilon) -> (loc, cov_diag, cov_factor) of hrnet_ssn (hrnet_module.py:559-595)."""
    size = x.shape[-2:]
    x = F.relu(_bn(_conv(x, sd, "conv1", stride=2), sd, "bn1"))
    x = F.relu(_bn(_conv(x, sd, "conv2", stride=2), sd, "bn2"))
    s1 = extra["STAGE1"]
    for b in range(s1["NUM_BLOCKS"][0]):
        x = _block(x, sd, f"layer1.{b}", s1["BLOCK"])
    ys = [x]
    n_prev = 1
    for si, tname in ((2, "transition1"), (3, "transition2"), (4, "transition3")):
        cfg = extra[f"STAGE{si}"]
        nb = cfg["NUM_BRANCHES"]
        xs = _transition(ys, sd, tname, n_prev, nb)
        for m in range(cfg["NUM_MODULES"]):
            xs = _module(xs, sd, f"stage{si}.{m}", nb, cfg["NUM_BLOCKS"], cfg["BLOCK"])
        ys, n_prev = xs, nb
    feats = list(ys)
    if extra.get("DROPOUT_FINAL", False) and dropout_masks is not None:
        feats = [f * m.to(f.dtype) * 2.0 for f, m in zip(feats, dropout_masks)]
    h, w = feats[0].shape[-2:]
    ups = [feats[0]] + [F.interpolate(f, size=(h, w), mode="bilinear", align_corners=ALIGN_CORNERS) for f in feats[1:]]
    cat = torch.cat(ups, 1)
    if ssn is not None:
        return _ssn_head(cat, sd, size, *ssn)
    y = F.relu(_bn(_conv(cat, sd, "last_layer.0"), sd, "last_layer.1"))
    y = _conv(y, sd, "last_layer.3", pad=1 if extra["FINAL_CONV_KERNEL"] == 3 else 0)
    y = F.interpolate(y, size=size, mode="bilinear", align_corners=ALIGN_CORNERS)
    if return_features:
        return y, ys
    return y


def _ssn_head(cat, sd, size, num_classes, rank, epsilon):
    """hrnet_module.py:559-583: mean and cov_diag BOTH come from last_layer (the reference calls it twice), the factor
    from cov_factor_conv; each is interpolated to the input size and flattened as the reference does."""
    def last(x, head):
        y = F.relu(_bn(_conv(x, sd, head + ".0"), sd, head + ".1"))
        return _conv(y, sd, head + ".3", pad=0)
    b = cat.shape[0]
    mean = F.interpolate(last(cat, "last_layer"), size=size, mode="bilinear", align_corners=ALIGN_CORNERS).reshape(b, -1)
    diag = F.interpolate(last(cat, "last_layer").exp() + epsilon, size=size, mode="bilinear",
                         align_corners=ALIGN_CORNERS).reshape(b, -1)
    fac = F.interpolate(last(cat, "cov_factor_conv"), size=size, mode="bilinear", align_corners=ALIGN_CORNERS)
    fac = fac.reshape(b, rank, num_classes, -1).flatten(2, 3).transpose(1, 2)
    return mean, diag, fac
