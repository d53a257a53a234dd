"""VisualEncoder / AudioEncoder with the reference's constructor and forward signatures (model/encoder.py:57-100),
running on the HIP kernels of libavhip.so.  There is no PyTorch-op fallback: on a box without the library or
without a GPU the forward raises.
"""
from __future__ import annotations

import os
from typing import Optional, Union

import torch
import torch.nn as nn

from ..utils import init as _init
from .w2v2 import Wav2Vec2ModelHIP, load_local_config, load_local_weights, w2v2_apply


class AudioEncoder(nn.Module):
    """model/encoder.py:80-100.  ``model_name`` is a LOCAL HF wav2vec2 directory (config.json [+ model.safetensors]);
    a dict is accepted as an in-memory config (random init).  Returns (last_hidden_state, mean(hidden_states[6:10]))."""

    def __init__(self, model_name: Union[str, dict] = "kresnik/wav2vec2-large-xlsr-korean", freeze: bool = True, seed: int = 2,
                 random_init: bool = False):
        super().__init__()
        weights = None
        if isinstance(model_name, dict):                      # in-memory config: seeded random weights (tests, benchmarks)
            cfg = model_name
        elif os.path.isdir(model_name):
            cfg = load_local_config(model_name)
            weights = load_local_weights(model_name)
            if weights is None and not random_init:
                raise FileNotFoundError(
                    f"AudioEncoder: {model_name!r} holds neither model.safetensors nor pytorch_model.bin; training would start from an "
                    "untrained wav2vec2 — pass random_init=True if that is intended")
        else:
            raise FileNotFoundError(
                f"AudioEncoder: {model_name!r} is not a local directory; pretrained checkpoints cannot be fetched "
                "(no network) — pass a local HF wav2vec2 directory or a config dict")
        self.model = Wav2Vec2ModelHIP(cfg)
        if weights is not None and not cfg.get("conv_bias", True):
            # conv_bias=False checkpoints have no feature-extractor conv biases: the kernels always add one, so they are zeros
            for i, c in enumerate(cfg["conv_dim"]):
                weights.setdefault(f"feature_extractor.conv_layers.{i}.conv.bias", torch.zeros(c))
        sd = _init.w2v2_state_dict(cfg, seed=seed, prefix="") if weights is None else weights
        res = self.model.load_state_dict(sd, strict=False)
        if weights is not None:
            # masked_spec_embed is absent from checkpoints saved with both SpecAugment probabilities at 0 (hf:1252-1253)
            missing = [k for k in res.missing_keys if k != "masked_spec_embed"]
            if missing or res.unexpected_keys:
                raise KeyError(f"AudioEncoder: checkpoint does not match the architecture of config.json: missing {missing[:5]}, "
                               f"unexpected {list(res.unexpected_keys)[:5]}")
        self.output_dim = self.model.config.hidden_size
        if freeze:
            for p in self.model.parameters():
                p.requires_grad = False

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, valid_lengths=None):
        """``valid_lengths`` (extension, optional): attention_mask.sum(-1) as host ints, when the caller has them (no device read-back)."""
        if not x.is_cuda:
            raise RuntimeError("AudioEncoder (HIP): input must be on the GPU; there is no CPU fallback")
        return w2v2_apply(self.model, x, attention_mask, valid_lengths=None if valid_lengths is None else (valid_lengths,))

    def forward_pair(self, x: torch.Tensor, attention_mask1: Optional[torch.Tensor], attention_mask2: Optional[torch.Tensor],
                     valid_lengths1=None, valid_lengths2=None):
        """The reference's two calls ``audio_encoder(audio, mask1)`` / ``audio_encoder(audio, mask2)`` (model/trainer.py:94-95) as one
        autograd node -> (last1, mid1, last2, mid2): same values, but the two backward passes run layer by layer with their weight
        gradients accumulated in place (one gradient bucket per layer and step under data parallelism)."""
        if not x.is_cuda:
            raise RuntimeError("AudioEncoder (HIP): input must be on the GPU; there is no CPU fallback")
        vl = None if valid_lengths1 is None or valid_lengths2 is None else (valid_lengths1, valid_lengths2)
        return w2v2_apply(self.model, x, attention_mask1, attention_mask2, two_passes=True, valid_lengths=vl)


# ---------------------------------------------------------------------------------------------------------------
# Visual encoder (model/encoder.py:6-75): Conv3d front-end + ResNet-18 trunk, BatchNorm + PReLU, applied per frame.
# torch.nn modules below are parameter/buffer CONTAINERS that reproduce the checkpoint keys
# (frontend3D.{0,1,2}.*, trunk.layerN.M.{conv1,bn1,relu,conv2,bn2,downsample.0,downsample.1}.*); the arithmetic
# runs in VisualEncoder.forward on implicit-GEMM MFMA convolutions over channel-last (NHWC) activations.
# ---------------------------------------------------------------------------------------------------------------
from .. import _lib as L          # noqa: E402
from .. import ops                # noqa: E402
from ..precision import compute_dtype, is_lp  # noqa: E402

_TRUNK = ((64, 1), (128, 2), (256, 2), (512, 2))      # (planes, stride of the first block) for layer1..4


FAST_C64 = os.environ.get("AVAMD_CONV_C64", "1") != "0"        # 0: layer1 through the implicit-GEMM kernel (A/B runs)
FUSE_MID_ACT = os.environ.get("AVAMD_FUSE_MID_ACT", "1") != "0"  # 0: separate BN-apply + PReLU pass between conv1 and conv2 of layer1 (A/B runs)
# ResNet layers 2-4 in POSITION-MAJOR pixel order (blocks of 256 frames, [block][y][x][frame of the block][C] instead of [frame][y][x][C], bf16 mode, frame count a
# multiple of 256): every row tile of a convolution then
# lies at one image position, and the filter taps that fall outside the (12x12 / 6x6 / 3x3) image for that position are skipped as whole
# K-tiles instead of multiplying zero lines - 11 % / 21 % / 40 % of the implicit-GEMM work of layer2 / layer3 / layer4.  BatchNorm, PReLU and
# the residual adds are per-pixel, so they do not see the order; the average pool reads it.  0: frame-major everywhere (A/B runs)
POS_MAJOR = os.environ.get("AVAMD_CONV_POSMAJOR", "1") != "0"
# Front-end: Conv3d fused with the window max / min of its raw output (frontend3d.hip POOL form + av_bn_prelu_minmax): the 1.9 GB conv output of a
# 64 x 100-frame batch is neither written nor read back; bit-identical to the unfused path.  0: conv -> HBM -> BN + PReLU + MaxPool pass (A/B runs)
FRONT_POOL = os.environ.get("AVAMD_FRONT_POOL", "1") != "0"


class BasicBlock(nn.Module):
    """Container for one residual block: conv3x3-BN-PReLU-conv3x3-BN (+1x1 conv-BN shortcut) -> add -> PReLU.
    A single PReLU (``relu``) serves both activations, as in the reference (model/encoder.py:11,18,22)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="prelu"):
        super().__init__()
        if relu_type != "prelu":
            raise NotImplementedError("the HIP visual encoder implements relu_type='prelu' (what main.py:92 uses)")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.PReLU(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), relu_type="prelu"):
        super().__init__()
        inpl = 64
        for li, ((planes, stride), nblk) in enumerate(zip(_TRUNK, layers), start=1):
            blocks = []
            for bi in range(nblk):
                st = stride if bi == 0 else 1
                ds = None
                if st != 1 or inpl != planes:
                    ds = nn.Sequential(nn.Conv2d(inpl, planes, 1, st, bias=False), nn.BatchNorm2d(planes))
                blocks.append(block(inpl, planes, st, ds, relu_type))
                inpl = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))


class VisualEncoder(nn.Module):
    def __init__(self, relu_type="prelu"):
        super().__init__()
        if relu_type != "prelu":
            raise NotImplementedError("the HIP visual encoder implements relu_type='prelu'")
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(64), nn.PReLU(64), nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        self.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type=relu_type)
        self.output_dim = 512
        self._wcache = {}
        self._nbt = []
        self._bn_ws = None

    # conv weight [Cout,Cin,kh,kw] -> [Cout, kh*kw*Cin] (tap-major K = the NHWC im2col order), compute dtype, cached
    def _w(self, conv: nn.Module, dtype):
        p = conv.weight
        key = id(p)
        ver = (p._version, p.data_ptr(), dtype)
        hit = self._wcache.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                if p.dim() == 5:
                    w = p.data.reshape(p.shape[0], -1)
                else:
                    w = p.data.permute(0, 2, 3, 1).reshape(p.shape[0], -1)
                hit = (ver, ops.cast(w.contiguous(), dtype))
            self._wcache[key] = hit
        return hit[1]

    def _w_front(self, conv: nn.Module):
        """[64,1,5,7,7] -> bf16 [64][288]: k = (kt*7+ky)*8 + kx, kx padded 7->8 and K padded 280->288 with zeros."""
        p = conv.weight
        key = ("front", id(p))
        ver = (p._version, p.data_ptr())
        hit = self._wcache.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                w = torch.zeros((64, 36, 8), dtype=torch.float32, device=p.device)
                w[:, :35, :7] = p.data.reshape(64, 35, 7)
                hit = (ver, ops.cast(w.reshape(64, 288).contiguous(), compute_dtype()))
            self._wcache[key] = hit
        return hit[1]

    def warm_caches(self, dtype) -> None:
        """Build (or refresh) every cached weight re-layout on the CURRENT stream.  The trainer runs the two lip streams on two HIP streams;
        called before they fork, so that the following stream never reads a re-layout the leading stream is still writing (first step)."""
        conv0 = self.frontend3D[0]
        if is_lp(dtype) and tuple(conv0.kernel_size) == (5, 7, 7):
            self._w_front(conv0)
        self._w(conv0, dtype)
        for li in range(1, 5):
            for blk in getattr(self.trunk, f"layer{li}"):
                self._w(blk.conv1, dtype); self._w(blk.conv2, dtype)
                if blk.downsample is not None:
                    self._w(blk.downsample[0], dtype)

    def _bn(self, bn: nn.Module, stats, nblk: int, count: int, training: bool):
        C = bn.num_features
        dev = bn.weight.device
        scale = torch.empty(C, dtype=torch.float32, device=dev); shift = torch.empty(C, dtype=torch.float32, device=dev)
        ws = None
        if training:                                          # one zeroed workspace per encoder AND stream: the finalize kernel leaves it zeroed
            sid = ops.stream()
            if self._bn_ws is None or not isinstance(self._bn_ws, dict):
                self._bn_ws = {}
            ws = self._bn_ws.get(sid)
            if ws is None or ws.device != dev:
                ws = self._bn_ws[sid] = torch.zeros(2 * 1024, dtype=torch.float64, device=dev)
            # two forward calls of one step on two streams (the trainer's two lip streams): the running statistics must be updated in call
            # order (model/trainer.py:88-89: lip1, then lip2), BatchNorm by BatchNorm - the leading call records an event behind each of
            # its finalize kernels, the following call waits for it
            sync = getattr(self, "_bn_sync", None)
            if sync is not None:
                mode, evs = sync
                i = self._bn_idx
                self._bn_idx += 1
                if mode == "follow" and i < len(evs):
                    torch.cuda.current_stream(dev).wait_event(evs[i])
        L.check(L.lib().av_bn_finalize(ops.ptr(stats), nblk, count, ops.ptr(bn.weight.data), ops.ptr(bn.bias.data),
                                       ops.ptr(bn.running_mean), ops.ptr(bn.running_var), float(bn.momentum), float(bn.eps),
                                       int(training), ops.ptr(scale), ops.ptr(shift), C, ops.ptr(ws), 1, ops.stream()), "av_bn_finalize")
        if training and getattr(self, "_bn_sync", None) is not None and self._bn_sync[0] == "lead":
            ev = torch.cuda.Event(); ev.record()
            self._bn_sync[1].append(ev)
        if training:
            self._nbt.append(bn.num_batches_tracked)        # bumped once per forward with one multi-tensor add (27 tiny launches otherwise)
        return scale, shift

    def _c64_ok(self, conv: nn.Module, dtype, N, H, W, Cin) -> bool:
        k, st, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        return (is_lp(dtype) and (k, st, pad, Cin, conv.out_channels) == (3, 1, 1, 64, 64) and W <= 31
                and N * H * W < (1 << 24) and FAST_C64)

    def _conv2d(self, x, N, H, W, Cin, conv: nn.Module, dtype, training: bool, in_act=None, pm: int = 0):
        """``in_act`` = (scale, shift, slope): x is the RAW output of the previous convolution and its BatchNorm-apply + PReLU is
        applied while the input window is staged (layer1 kernel only: callers check ``_c64_ok``).
        ``pm``: bit 0 = x is in position-major pixel order, bit 1 = write the output position-major (av_gemm_args.cPM)."""
        k, st = conv.kernel_size[0], conv.stride[0]
        pad = conv.padding[0]
        Cout = conv.out_channels
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        M = N * Ho * Wo
        y = torch.empty((M, Cout), dtype=dtype, device=x.device)
        if self._c64_ok(conv, dtype, N, H, W, Cin):
            # layer1: weights-stationary kernel (conv3x3_c64.hip), one BN partial per 256 output pixels
            nblk = (M + 255) // 256
            stats = torch.empty((nblk, 2, Cout), dtype=torch.float32, device=x.device) if training else None
            isc, ish, isl = in_act if in_act is not None else (None, None, None)
            L.check(L.lib().av_conv3x3_c64(ops.ptr(x), ops.ptr(self._w(conv, dtype)), ops.ptr(y), ops.ptr(stats), N, H, W,
                                           ops.ptr(isc), ops.ptr(ish), ops.ptr(isl), ops.stream()), "av_conv3x3_c64")
            return y, stats, nblk, M, Ho, Wo
        assert in_act is None
        nblk = (M + 127) // 128
        stats = torch.empty((nblk, 2, Cout), dtype=torch.float32, device=x.device) if training else None
        geo = dict(cT=1, cH=H, cW=W, cCtot=Cin, cCin=Cin, cCoff=0, cKt=1, cKh=k, cKw=k, cSh=st, cSw=st, cPt=0, cPh=pad, cPw=pad,
                   cOh=Ho, cOw=Wo, cNF=256 if pm else 0, cPM=pm)
        ops.gemm(x, self._w(conv, dtype), y, M=M, N=Cout, K=k * k * Cin, lda=0, ldb=k * k * Cin, ldc=Cout, a_mode=L.A_CONV2D,
                 conv=geo, stats=stats)
        return y, stats, nblk, M, Ho, Wo

    def _act(self, x, scale, shift, slope, res=None, rscale=None, rshift=None):
        out = torch.empty_like(x)
        L.check(L.lib().av_bn_act(ops.ptr(x), ops.ptr(scale), ops.ptr(shift), ops.ptr(res), ops.ptr(rscale), ops.ptr(rshift),
                                  ops.ptr(slope), ops.ptr(out), ops.dt(x), x.numel(), x.shape[-1], ops.stream()), "av_bn_act")
        return out

    @torch.no_grad()
    def _forward_impl(self, x: torch.Tensor) -> torch.Tensor:
        self._nbt = []
        if isinstance(self._bn_ws, dict) and x.is_cuda:
            ws = self._bn_ws.get(ops.stream())
            if ws is not None:
                ws.zero_()                                       # one clear per forward (robust against an aborted previous forward)
        dtype = compute_dtype()
        training = self.training            # .train() on the frozen encoder => batch statistics + running-stat update
        B, C, T, H, W = x.shape
        assert C == 1
        dev = x.device
        conv0 = self.frontend3D[0]
        kt, kh, kw = conv0.kernel_size
        Ho, Wo = (H + 2 * conv0.padding[1] - kh) // conv0.stride[1] + 1, (W + 2 * conv0.padding[2] - kw) // conv0.stride[2] + 1
        M = B * T * Ho * Wo
        fast = (is_lp(dtype) and (kt, kh, kw) == (5, 7, 7) and tuple(conv0.stride) == (1, 2, 2)
                and tuple(conv0.padding) == (2, 3, 3) and H % 16 == 0 and W % 32 == 0)
        N = B * T
        Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        pooled = None
        if fast and FRONT_POOL:
            # conv + window max / min of the raw output in one kernel (frontend3d.hip, POOL form): the 4x larger conv output never reaches HBM;
            # BatchNorm + PReLU + MaxPool (model/encoder.py:62-64) are finished from the two extremes once the batch statistics exist
            nblk = B * T * (Ho // 8) * (Wo // 16)
            stats = torch.empty((nblk, 2, 64), dtype=torch.float32, device=dev) if training else None
            pooled = torch.empty((2, N * Hp * Wp, 64), dtype=dtype, device=dev)
            L.check(L.lib().av_conv3d_front_pool(ops.ptr(x.contiguous().float()), ops.ptr(self._w_front(conv0)), ops.ptr(pooled[0]), ops.ptr(pooled[1]),
                                                 ops.ptr(stats), B, T, H, W, ops.stream()), "av_conv3d_front_pool")
        elif fast:    # patch-in-LDS implicit GEMM (frontend3d.hip); one BN partial per 8x16 output tile
            y = torch.empty((M, 64), dtype=dtype, device=dev)
            nblk = B * T * (Ho // 8) * (Wo // 16)
            stats = torch.empty((nblk, 2, 64), dtype=torch.float32, device=dev) if training else None
            L.check(L.lib().av_conv3d_front(ops.ptr(x.contiguous().float()), ops.ptr(self._w_front(conv0)), ops.ptr(y), ops.ptr(stats),
                                            B, T, H, W, ops.stream()), "av_conv3d_front")
        else:
            y = torch.empty((M, 64), dtype=dtype, device=dev)
            xin = ops.cast(x.contiguous().float().view(B, T, H, W), dtype)
            nblk = (M + 127) // 128
            stats = torch.empty((nblk, 2, 64), dtype=torch.float32, device=dev) if training else None
            geo = dict(cT=T, cH=H, cW=W, cCtot=1, cCin=1, cCoff=0, cKt=kt, cKh=kh, cKw=kw, cSh=conv0.stride[1], cSw=conv0.stride[2],
                       cPt=conv0.padding[0], cPh=conv0.padding[1], cPw=conv0.padding[2], cOh=Ho, cOw=Wo)
            ops.gemm(xin, self._w(conv0, dtype), y, M=M, N=64, K=kt * kh * kw, lda=0, ldb=kt * kh * kw, ldc=64, a_mode=L.A_CONV3D1,
                     conv=geo, stats=stats)
        sc, sh = self._bn(self.frontend3D[1], stats, nblk, M, training)
        h = torch.empty((N * Hp * Wp, 64), dtype=dtype, device=dev)
        if pooled is not None:
            L.check(L.lib().av_bn_prelu_minmax(ops.ptr(pooled[0]), ops.ptr(pooled[1]), ops.ptr(sc), ops.ptr(sh), ops.ptr(self.frontend3D[2].weight.data),
                                               ops.ptr(h), h.numel(), ops.stream()), "av_bn_prelu_minmax")
        else:
            L.check(L.lib().av_bn_prelu_maxpool(ops.ptr(y), ops.ptr(sc), ops.ptr(sh), ops.ptr(self.frontend3D[2].weight.data), ops.ptr(h),
                                                ops.dt(h), N, Ho, Wo, 64, ops.stream()), "av_bn_prelu_maxpool")
        Hc, Wc, Cc = Hp, Wp, 64
        # position-major order from layer2 on (bf16 fast path: input channels a multiple of 64, plain 3x3 / 1x1 filters)
        FB = 256                                                # frames per position-major block = the row tile of the 8-phase kernel
        use_pm = POS_MAJOR and is_lp(dtype) and N % FB == 0
        pm_in = 0                                               # pixel order of h
        for li in range(1, 5):
            for blk in getattr(self.trunk, f"layer{li}"):
                slope = blk.relu.weight.data
                pm_out = 1 if (use_pm and li >= 2) else 0
                pm1 = pm_in | (pm_out << 1)                     # conv1 / downsample: read h's order, write the layer's order
                pm2 = pm_out * 3                                # conv2: both sides in the layer's order
                if pm_out:
                    c1, st1, nb1, M1, H1, W1 = self._conv2d(h, N, Hc, Wc, Cc, blk.conv1, dtype, training, pm=pm1)
                    s1, b1 = self._bn(blk.bn1, st1, nb1, M1, training)
                    a1 = self._act(c1, s1, b1, slope)
                    c2, st2, nb2, M2, _, _ = self._conv2d(a1, N, H1, W1, blk.conv1.out_channels, blk.conv2, dtype, training, pm=pm2)
                    s2, b2 = self._bn(blk.bn2, st2, nb2, M2, training)
                    if blk.downsample is not None:
                        cd, std, nbd, Md, _, _ = self._conv2d(h, N, Hc, Wc, Cc, blk.downsample[0], dtype, training, pm=pm1)
                        sd, bd = self._bn(blk.downsample[1], std, nbd, Md, training)
                        h = self._act(c2, s2, b2, slope, res=cd, rscale=sd, rshift=bd)
                    else:
                        # the residual is added row by row: h must already be in the layer's (position-major) order.  True for ResNet-18, where the
                        # first block of layer2 has a downsample branch; any other trunk would silently add permuted rows
                        assert pm_in == pm_out, "position-major trunk: an identity residual needs its input in the layer's pixel order"
                        h = self._act(c2, s2, b2, slope, res=h)
                    Hc, Wc, Cc = H1, W1, blk.conv1.out_channels
                    pm_in = 1
                    continue
                c1, st1, nb1, M1, H1, W1 = self._conv2d(h, N, Hc, Wc, Cc, blk.conv1, dtype, training)
                s1, b1 = self._bn(blk.bn1, st1, nb1, M1, training)
                if FUSE_MID_ACT and self._c64_ok(blk.conv2, dtype, N, H1, W1, blk.conv1.out_channels):
                    # mid-block BN-apply + PReLU inside conv2's window staging: no HBM pass for the activated tensor
                    c2, st2, nb2, M2, _, _ = self._conv2d(c1, N, H1, W1, blk.conv1.out_channels, blk.conv2, dtype, training,
                                                          in_act=(s1, b1, slope))
                else:
                    a1 = self._act(c1, s1, b1, slope)
                    c2, st2, nb2, M2, _, _ = self._conv2d(a1, N, H1, W1, blk.conv1.out_channels, blk.conv2, dtype, training)
                s2, b2 = self._bn(blk.bn2, st2, nb2, M2, training)
                if blk.downsample is not None:
                    cd, std, nbd, Md, _, _ = self._conv2d(h, N, Hc, Wc, Cc, blk.downsample[0], dtype, training)
                    sd, bd = self._bn(blk.downsample[1], std, nbd, Md, training)
                    h = self._act(c2, s2, b2, slope, res=cd, rscale=sd, rshift=bd)
                else:
                    h = self._act(c2, s2, b2, slope, res=h)
                Hc, Wc, Cc = H1, W1, blk.conv1.out_channels
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
            self._nbt = []
        out = torch.empty((N, Cc), dtype=torch.float32, device=dev)
        L.check(L.lib().av_avgpool(ops.ptr(h), ops.dt(h), ops.ptr(out), N, Hc * Wc, Cc, FB if pm_in else 0, ops.stream()), "av_avgpool")
        return out.view(B, T, Cc)

    def forward(self, x):
        """x [B,1,T,96,96] -> [B,T,512] (model/encoder.py:69-75)."""
        if not x.is_cuda:
            raise RuntimeError("VisualEncoder (HIP): input must be on the GPU; there is no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the HIP visual encoder is forward-only: freeze trunk and frontend3D as main.py:100-103 does")
        return self._forward_impl(x)
