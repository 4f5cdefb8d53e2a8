"""UNet3D on MI355X: the reference's model class surface, HIP kernels underneath.

Mirrors `uncertainty_modeling.models.unet3D_module.UNet3D` (unet3D_module.py:7-373): identical
constructor keywords, identical state-dict key names (`contr_1_1.0.weight`, `center.4.bias`,
`upscale3.weight`, `final.weight`, ...), so a hydra `_target_` can be re-pointed to
`values_amd.unet3d.UNet3D` and the same Lightning checkpoint loads through
`load_state_dict` (test_3D.py:222-247).  `forward(x)` takes the reference's (N,1,D,H,W) tensor and
returns (N,C,D,H,W) logits -- computed by libvalues_amd.so (vx_unet3d_forward), never by ATen.

The torch.nn layers created here are PARAMETER CONTAINERS only (they give load_state_dict / .to() /
.double() for free); their own forward is never called.

Extensions used by the batched multi-pass driver (values_amd.predict):
    forward(x, n_samples=T)   T MC-dropout samples per volume in one launch (sample n reads volume n // T)
    forward(x, src=, flip=)   per-sample source volume and TTA flip code (bit0 = dim 2, bit1 = dim 3, bit2 = dim 4)
    forward(x, dropout_masks=[17 bool tensors])   inject the reference's keep-masks (parity tests)
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib

# execution order of the 18 3x3x3 convs, the 4 transposed convs and the 17 dropouts
CONV_ORDER = ["contr_1_1.0", "contr_1_2.0", "contr_2_1.0", "contr_2_2.0", "contr_3_1.0", "contr_3_2.0",
              "contr_4_1.0", "contr_4_2.0", "center.0", "center.2", "expand_4_1.0", "expand_4_2.0",
              "expand_3_1.0", "expand_3_2.0", "expand_2_1.0", "expand_2_2.0", "expand_1_1.0", "expand_1_2.0"]
UP_ORDER = ["center.4", "upscale4", "upscale3", "upscale2"]
DROPOUT_ORDER = ["contr_1_1", "contr_1_2", "contr_2_1", "contr_2_2", "contr_3_1", "contr_3_2", "contr_4_1",
                 "contr_4_2", "center", "expand_4_1", "expand_4_2", "expand_3_1", "expand_3_2", "expand_2_1",
                 "expand_2_2", "expand_1_1", "expand_1_2"]


def _block(cin, cout, k, norm, p):
    layers = [nn.Conv3d(cin, cout, k, padding=1)]
    if norm:
        layers.append(nn.InstanceNorm3d(cout))
    layers += [nn.LeakyReLU(inplace=True), nn.Dropout(p=p)]
    return nn.Sequential(*layers)


class UNet3D(nn.Module):
    def __init__(self, num_classes: int, in_channels: int = 1, initial_filter_size: int = 8, kernel_size: int = 3,
                 do_instancenorm: bool = True, do_dropout: bool = False, aleatoric_loss: bool = False):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("values_amd.UNet3D: only kernel_size=3 (every shipped config) has a HIP kernel")
        if not 1 <= in_channels <= 8:
            raise NotImplementedError("values_amd.UNet3D: in_channels must be 1 .. 8")
        if initial_filter_size not in (8, 16, 32):
            raise NotImplementedError("values_amd.UNet3D: initial_filter_size must be 8, 16 or 32")
        self.num_classes = num_classes
        self.aleatoric_loss = aleatoric_loss
        self.in_channels = in_channels
        self.do_instancenorm = bool(do_instancenorm)
        self.initial_filter_size = initial_filter_size
        self.dropout_prob = 0.5 if do_dropout else 0.0  # unet3D_module.py:31-34
        f, p = initial_filter_size, self.dropout_prob
        chans = [f, 2 * f, 4 * f, 8 * f]
        prev = in_channels
        for lvl, c in enumerate(chans, start=1):
            setattr(self, f"contr_{lvl}_1", _block(prev, c, 3, do_instancenorm, p))
            setattr(self, f"contr_{lvl}_2", _block(c, c, 3, do_instancenorm, p))
            prev = c
        center = [nn.Conv3d(8 * f, 16 * f, 3, padding=1), nn.ReLU(inplace=True),
                  nn.Conv3d(16 * f, 16 * f, 3, padding=1), nn.ReLU(inplace=True),
                  nn.ConvTranspose3d(16 * f, 8 * f, 2, stride=2), nn.ReLU(inplace=True)]
        if do_dropout:
            center.append(nn.Dropout(p=p))
        self.center = nn.Sequential(*center)
        for lvl in (4, 3, 2, 1):
            c = chans[lvl - 1]
            setattr(self, f"expand_{lvl}_1", _block(2 * c, c, 3, False, p))
            setattr(self, f"expand_{lvl}_2", _block(c, c, 3, False, p))
            if lvl > 1:
                setattr(self, f"upscale{lvl}", nn.ConvTranspose3d(c, c // 2, kernel_size=2, stride=2))
        self.final = nn.Conv3d(f, num_classes, kernel_size=1)
        if aleatoric_loss:
            self.final_aleatoric = nn.Conv3d(f, num_classes * 2, kernel_size=1)
        self.output_reconstruction_map = nn.Conv3d(f, out_channels=1, kernel_size=1)
        self._packed = None
        self._packed_key = None
        self._packed_by_mode = {}   # (pack_mode(), device) -> (parameter key, (vx_unet3d_weights, tensors it points into))
        self._range = {}      # device -> int32 word: running maximum |activation| handed to a split-fp16 conv (bit pattern)
        self._ws = {}
        self._calls = 0
        self.seed = 123  # reference default seed (configs/dropout_config.yaml:8); set_seed analogue

    # ------------------------------------------------------------------ weights
    def _param_key(self):
        return tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) for p in self.parameters()) + (_lib.pack_mode(),)

    def _ensure_packed(self, device):
        """Packed weights of the CURRENT kernel family (vx_config's pack_mode()).  One entry per family stays alive: a
        switch of the family (the fp16-range fallback runs a batch under conv_fp32 = 1) must not free the tensors of the
        other one -- a captured hipGraph (GraphedPredictor) replays with the pointers it was captured with.  An entry is
        replaced only when the parameters themselves changed."""
        key = self._param_key()
        mode = (key[-1], str(device))
        hit = self._packed_by_mode.get(mode)
        if hit is not None and hit[0] == key:
            self._packed, self._packed_key = hit[1], key
            return self._packed
        lib = _lib.load()
        sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in self.state_dict().items()}
        keep = []  # tensors the ctypes struct points into
        w = _lib.UNet3DWeights()
        st = _lib.stream_ptr()
        for i, name in enumerate(CONV_ORDER):
            wt, b = sd[name + ".weight"], sd[name + ".bias"]
            cout, cin = wt.shape[0], wt.shape[1]
            if i == 0 and cin > 1:
                # in_channels > 1: zero-padded to 8 input channels for the general kernels (vx_pack_input_cl8 pads the input)
                wt8 = torch.zeros((cout, 8) + tuple(wt.shape[2:]), dtype=torch.float32, device=device)
                wt8[:, :cin] = wt
                wt, cin = wt8.contiguous(), 8
                n = lib.vx_conv3d_k3_packed_floats(cin, cout)
                packed = torch.empty(n, dtype=torch.float32, device=device)
                _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wt), _lib.ptr(packed), cin, cout, st), "vx_pack_conv3d_k3")
            elif i == 0:
                packed = wt  # Cin == 1 kernel reads the torch layout
            else:
                n = lib.vx_conv3d_k3_packed_floats(cin, cout)
                if n < 0:
                    raise _lib.VxError(f"conv {name}: Cin={cin} Cout={cout} unsupported")
                packed = torch.empty(n, dtype=torch.float32, device=device)
                _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wt), _lib.ptr(packed), cin, cout, st), "vx_pack_conv3d_k3")
            keep += [packed, b, wt]
            w.conv_w[i] = packed.data_ptr()
            w.conv_b[i] = b.data_ptr()
            w.conv_family[i] = 0 if (i == 0 and cin == 1) else lib.vx_conv3d_k3_family(cin, cout)
        for i, name in enumerate(UP_ORDER):
            wt, b = sd[name + ".weight"], sd[name + ".bias"]
            cin, cout = wt.shape[0], wt.shape[1]
            n = lib.vx_convT_k2s2_packed_floats(cin, cout)
            packed = torch.empty(n, dtype=torch.float32, device=device)
            _lib.check(lib.vx_pack_convT_k2s2(_lib.ptr(wt), _lib.ptr(packed), cin, cout, st), "vx_pack_convT_k2s2")
            keep += [packed, b, wt]
            w.up_w[i] = packed.data_ptr()
            w.up_b[i] = b.data_ptr()
        # the level-0 up-convolution composed into expand_1_1's weights (vx_conv3d_args.up_fused): F = 8 networks on the
        # split-fp16 family (the z-column kernel that takes it)
        if self.initial_filter_size == 8 and _lib.get_config().conv_fp32 == 0:
            uf = torch.empty(lib.vx_conv3d_upfused_packed_floats(), dtype=torch.float32, device=device)
            _lib.check(lib.vx_pack_conv3d_upfused(_lib.ptr(sd["expand_1_1.0.weight"]), _lib.ptr(sd["expand_1_1.0.bias"]),
                                                  _lib.ptr(sd["upscale2.weight"]), _lib.ptr(sd["upscale2.bias"]), _lib.ptr(uf), st),
                       "vx_pack_conv3d_upfused")
            keep.append(uf)
            w.up_fused = uf.data_ptr()
            # (round 5) expand_2_1 as two 16 -> 16 convs over the halves of its input (vx_unet3d_weights.split_w)
            w2 = sd["expand_2_1.0.weight"]
            for hidx in range(2):
                part = w2[:, 16 * hidx:16 * (hidx + 1)].contiguous()
                pk = torch.empty(lib.vx_conv3d_k3_packed_floats(16, 16), dtype=torch.float32, device=device)
                _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(part), _lib.ptr(pk), 16, 16, st), "vx_pack_conv3d_k3 (expand_2_1 half)")
                keep += [part, pk]
                w.split_w[hidx] = pk.data_ptr()
            w.split_family = lib.vx_conv3d_k3_family(16, 16)
            u3 = torch.empty(lib.vx_convT_zc16_packed_floats(), dtype=torch.float32, device=device)
            _lib.check(lib.vx_pack_convT_zc16(_lib.ptr(sd["upscale3.weight"]), _lib.ptr(u3), st), "vx_pack_convT_zc16")
            keep.append(u3)
            w.up3_zc16 = u3.data_ptr()
        fw, fb = self._head_params(sd)
        keep += [fw, fb]
        w.final_w = fw.data_ptr()
        w.final_b = fb.data_ptr()
        w.F = self.initial_filter_size
        w.num_classes = fw.shape[0]
        w.in_channels = self.in_channels
        w.no_instancenorm = 0 if self.do_instancenorm else 1
        self._packed = (w, keep)
        self._packed_key = key
        self._packed_by_mode[mode] = (key, self._packed)
        return self._packed

    def _head_params(self, sd):
        """(weight (C, F), bias (C,)) of the 1x1x1 head the forward ends with (unet3D_module.py:199-204, 365-369)."""
        head = "final_aleatoric" if self.aleatoric_loss else "final"
        return sd[head + ".weight"].reshape(sd[head + ".weight"].shape[0], -1).contiguous(), sd[head + ".bias"]

    def _workspace(self, N, D, H, W, device):
        # one workspace per stream (forwards of volume chunks may be in flight on several streams), each keeping the
        # last geometry it was used with
        geo = (N, D, H, W, str(device))
        sid = torch.cuda.current_stream(device).cuda_stream
        ent = self._ws.get(sid)
        if ent is None or ent[0] != geo:
            nbytes = _lib.load().vx_unet3d_workspace_bytes(N, D, H, W, self.initial_filter_size)
            self._ws.pop(sid, None)
            ent = (geo, torch.empty(nbytes + 256, dtype=torch.uint8, device=device))
            self._ws[sid] = ent
        ws = ent[1]
        off = (-ws.data_ptr()) % 256
        return ws, off, ws.numel() - 256

    FP16_MAX = 65504.0

    def range_max(self, reset: bool = False) -> float:
        """Largest |activation| an un-normalised layer (center, decoder, transposed convs) has handed to a split-fp16
        convolution since the last reset, over all forwards of this model -- 0.0 while everything stayed below 32768 (a
        device word the kernels raise with an atomic max once a value gets within a factor two of the limit; reading it
        synchronises).  At or beyond 65504 the fp16 split of that value overflowed: the logits of that
        forward are invalid.  The native-fp32 kernels (vx_config.conv_fp32 = 1) have no such limit."""
        worst = 0.0
        for flag in self._range.values():
            worst = max(worst, float(flag.view(torch.float32).item()))
            if reset:
                flag.zero_()
        return worst

    def range_reset(self):
        """zero the range word(s) in stream order, WITHOUT reading (no synchronisation: pipelined callers)"""
        for flag in self._range.values():
            flag.zero_()

    def next_seed(self) -> int:
        """the hash-dropout seed the next un-seeded forward would draw; drawing it here pins it (a re-run of the same
        batch -- the fp16-range fallback -- then replays the same dropout bits)"""
        seed = (self.seed * 1000003 + self._calls) & 0xFFFFFFFF
        self._calls += 1
        return seed

    def check_range(self, reset: bool = True):
        m = self.range_max(reset=reset)
        if not m < self.FP16_MAX:        # also catches NaN
            raise _lib.VxError(f"values_amd.UNet3D: an activation of magnitude {m:.4g} reached a split-fp16 convolution "
                               f"(limit {self.FP16_MAX}); re-run under _lib.config(conv_fp32=1)")

    def dropout_layer_shapes(self, D, H, W):
        """[(channels, (d, h, w))] of the 17 dropout layers in DROPOUT_ORDER for a (D, H, W) input."""
        f = self.initial_filter_size
        enc = [(f << l, (D >> l, H >> l, W >> l)) for l in range(4) for _ in range(2)]
        dec = [(f << l, (D >> l, H >> l, W >> l)) for l in (3, 2, 1, 0) for _ in range(2)]
        return enc + [(8 * f, (D >> 3, H >> 3, W >> 3))] + dec

    def hash_dropout_masks(self, seed: int, N: int, D: int, H: int, W: int, device=None):
        """The keep-masks the hash bit generator applies to an N-sample forward launched with `seed`
        (vx_drop_hash_mask), as 17 bool tensors (N, C, d, h, w) in DROPOUT_ORDER: feeding them back through
        `dropout_masks=` -- or to a float64 restatement of the network -- replays that forward's dropout exactly."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        out = []
        for layer, (c, (d, h, w)) in enumerate(self.dropout_layer_shapes(D, H, W)):
            m = torch.empty((N, d, h, w, c), dtype=torch.uint8, device=dev)
            _lib.check(lib.vx_drop_hash_mask(int(seed) & 0xFFFFFFFF, layer, N, d * h * w * c, _lib.ptr(m),
                                             _lib.stream_ptr()), "vx_drop_hash_mask")
            out.append(m.permute(0, 4, 1, 2, 3).bool())
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, enable_concat: bool = True, last_layer: bool = True, *,
                n_samples: int = 1, src: Optional[torch.Tensor] = None, flip: Optional[torch.Tensor] = None,
                dst: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                dropout_masks: Optional[Sequence[torch.Tensor]] = None, seed: Optional[int] = None,
                seed_dev: Optional[torch.Tensor] = None):
        if not enable_concat or not last_layer:
            raise NotImplementedError("values_amd.UNet3D: autoencoder / feature modes are training-only and not on the HIP path")
        res = self._run(x, n_samples=n_samples, src=src, flip=flip, dst=dst, out=out, dropout_masks=dropout_masks, seed=seed,
                        seed_dev=seed_dev)
        if self.aleatoric_loss:
            mu, s = res.split(self.num_classes, 1)  # unet3D_module.py:367-369
            return mu, s
        return res

    def _run(self, x, *, n_samples=1, src=None, flip=None, dst=None, out=None, dropout_masks=None, seed=None, seed_dev=None):
        """The whole network through vx_unet3d_forward: (N, head channels, D, H, W) in x's dtype."""
        _lib.require_gpu()
        lib = _lib.load()
        if x.dim() != 5 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected (N,{self.in_channels},D,H,W) input, got {tuple(x.shape)}")
        in_dtype = x.dtype
        dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        xf = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        V, _, D, H, W = xf.shape
        if src is not None:
            N = int(src.numel())
        else:
            N = V * int(n_samples)
        w, _keep = self._ensure_packed(dev)
        NC = w.num_classes
        ws, off, ws_bytes = self._workspace(N, D, H, W, dev)
        if out is None:
            out = torch.empty((N, NC, D, H, W), dtype=torch.float32, device=dev)
        run = _lib.UNet3DRun()
        run.x = xf.data_ptr()
        run.N, run.D, run.H, run.W = N, D, H, W
        run.repeat = int(n_samples)
        hold = [xf, out, ws]
        for name, t in (("src", src), ("flip", flip), ("dst", dst)):
            if t is not None:
                t = t.to(device=dev, dtype=torch.int32).contiguous()
                hold.append(t)
                setattr(run, name, t.data_ptr())
        if dropout_masks is not None:
            if len(dropout_masks) != 17:
                raise ValueError("dropout_masks: need 17 masks in DROPOUT_ORDER")
            run.drop_mode = _lib.VX_DROP_MASK
            for i, m in enumerate(dropout_masks):
                # reference layout (N,C,D,H,W) bool -> channels-last uint8
                m = m.to(device=dev).permute(0, 2, 3, 4, 1).contiguous().to(torch.uint8)
                hold.append(m)
                run.masks[i] = m.data_ptr()
        elif self.training and self.dropout_prob > 0:
            run.drop_mode = _lib.VX_DROP_HASH
            if seed is None:
                seed = self.next_seed()
            run.seed = int(seed) & 0xFFFFFFFF
            if seed_dev is not None:   # a device word added to the seed by every kernel (hipGraph replays, GraphedPredictor)
                if seed_dev.dtype != torch.int32 or not seed_dev.is_cuda:
                    raise ValueError("seed_dev: an int32 device tensor of one element")
                hold.append(seed_dev)
                run.seed_dev = seed_dev.data_ptr()
        else:
            run.drop_mode = _lib.VX_DROP_NONE
        flag = self._range.get(str(dev))
        if flag is None:
            flag = self._range[str(dev)] = torch.zeros(1, dtype=torch.int32, device=dev)
        run.range_flag = flag.data_ptr()
        run.logits = out.data_ptr()
        run.workspace = ws.data_ptr() + off
        run.workspace_bytes = ws_bytes
        _lib.check(lib.vx_unet3d_forward(C.byref(w), C.byref(run), _lib.stream_ptr()), "vx_unet3d_forward")
        self._hold = hold  # keep inputs alive until the stream has consumed them
        return out if in_dtype == torch.float32 else out.to(in_dtype)
