"""Device-level front end: torch tensors in, torch tensors out, every call a C-ABI call.

PyTorch is used here only for device memory and streams (the plumbing); the arithmetic of
every method below runs in libspalign.so (hand-written HIP for gfx950).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import FmapDesc, SpalignError, check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _req(t, dtype, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise SpalignError('%s must be a contiguous CUDA tensor of dtype %s' % (what, dtype))
    return t


class Engine(object):
    """One spa_ctx (one GPU). Not thread safe; make one per process/GPU."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise SpalignError('no GPU visible: the superpixel-align hot path has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self._lib = _lib.lib()
        h = ctypes.c_void_p()
        check(self._lib.spa_ctx_create(self.device.index, ctypes.byref(h)))
        self._ctx = h

    def _s(self):
        """The launch stream: torch's current stream of THIS engine's device (a spa_ctx is bound to one
        device; torch's current device may be another one)."""
        return _stream(self.device)

    def close(self):
        if getattr(self, '_ctx', None):
            self._lib.spa_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ status
    def status(self):
        """Read and clear the latched device status bits (synchronises the stream)."""
        v = ctypes.c_uint32(0)
        check(self._lib.spa_status(self._ctx, ctypes.byref(v), self._s()))
        return v.value

    def status_peek_async(self, word=None):
        """Enqueue a copy of the latched status bits into a pinned one-element int32 tensor on the current stream
        (nothing is cleared or synchronised); check it with `raise_on_word` once the stream got that far."""
        if word is None:
            word = torch.zeros(1, dtype=torch.int32).pin_memory()
        check(self._lib.spa_status_peek_async(self._ctx, ctypes.c_void_p(word.data_ptr()), self._s()))
        return word

    def status_take_async(self, word=None):
        """status_peek_async followed, in stream order, by the clear of the latch: the word holds the bits of the work
        enqueued since the previous take (per batch in the drivers' asynchronous loop)."""
        if word is None:
            word = torch.zeros(1, dtype=torch.int32).pin_memory()
        check(self._lib.spa_status_take_async(self._ctx, ctypes.c_void_p(word.data_ptr()), self._s()))
        return word

    def raise_on_word(self, word, ignore=_lib.INFO_BITS):
        st = int(word.item()) & 0xffffffff
        self.last_info = st & _lib.INFO_BITS
        st &= ~ignore
        if st:
            self.status()                                            # clear the latch, then report
            msgs = [m for bit, m in _lib.STATUS_BITS.items() if st & bit]
            raise SpalignError('device status 0x%x: %s' % (st, '; '.join(msgs)))

    def raise_on_status(self, ignore=_lib.INFO_BITS):
        """Raise on latched error bits.  Informational bits (a starved SLIC seed, an oversize
        component: both handled as scikit-image handles them) are collected in `self.last_info`."""
        st = self.status()
        self.last_info = st & _lib.INFO_BITS
        st &= ~ignore
        if st:
            msgs = [m for bit, m in _lib.STATUS_BITS.items() if st & bit]
            raise SpalignError('device status 0x%x: %s' % (st, '; '.join(msgs)))

    def ws_generation(self):
        """re-allocations of the context's workspaces so far (a captured graph is valid while this does not change)"""
        return int(self._lib.spa_ws_generation(self._ctx))

    def debug_set(self, key, value):
        """diagnostic kernel-selection switches (include/spalign.h: spa_debug_set); key 1 = planes-in-LDS kernel for the narrow
        split-plane 3x3 layers (1, default) or the kernel it replaced (0)"""
        check(self._lib.spa_debug_set(self._ctx, int(key), int(value)))

    # ------------------------------------------------------------------ per-kernel timing
    def prof_enable(self, on=True):
        check(self._lib.spa_prof_enable(self._ctx, 1 if on else 0))
        self._prof_on = bool(on)

    def prof_is_on(self):
        return bool(getattr(self, '_prof_on', False))

    def prof_read(self):
        """{kernel name: (total ms, launches)} since prof_enable (synchronises)."""
        out = {}
        for slot in range(self._lib.spa_prof_slots()):
            ms, n = ctypes.c_double(0), ctypes.c_int(0)
            check(self._lib.spa_prof_read(self._ctx, slot, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[self._lib.spa_prof_name(slot).decode()] = (ms.value, n.value)
        return out

    # ------------------------------------------------------------------ DRN glue
    def drn_normalise(self, x, dtype=torch.float32):
        """(B,3,H,W) float32 0..255 -> normalised channels-last tensor of `dtype` (one pass)."""
        x = _req(x, torch.float32, 'x')
        B, C, H, W = x.shape
        assert C == 3
        out = torch.empty((B, 3, H, W), dtype=dtype, device=x.device, memory_format=torch.channels_last)
        mean = (ctypes.c_double * 3)(0.485, 0.456, 0.406)
        std = (ctypes.c_double * 3)(0.229, 0.224, 0.225)
        check(self._lib.spa_drn_normalise(self._ctx, _ptr(x), B, H, W, _ptr(out),
                                          0 if dtype == torch.float32 else 1, mean, std, self._s()))
        return out

    def drn_stem_d(self, x, w0, b0, w1p, b1, dtype=torch.float32, split=False, want_layer0=False):
        """DRN-D stem in one kernel: raw (B,3,H,W) float32 0..255 -> layer1 output (B,16,H,W) of
        `dtype` (float32 arithmetic) in channels-last storage.  w0 (16,147), w1p (16,144) in
        (n, ky, kx, c) order.  split (float32 only): the 16-bit matrix cores with two half-precision planes per operand
        (float32 accuracy, csrc/spa_stem.hip) instead of the float32 matrix instructions."""
        x = _req(x, torch.float32, 'x')
        B, C, H, W = x.shape
        assert C == 3
        for t, shape in ((w0, (16, 147)), (b0, (16,)), (w1p, (16, 144)), (b1, (16,))):
            _req(t, torch.float32, 'stem weights')
            assert tuple(t.shape) == shape
        out = torch.empty((B, 16, H, W), dtype=dtype, device=x.device, memory_format=torch.channels_last)
        mean = (ctypes.c_double * 3)(0.485, 0.456, 0.406)
        std = (ctypes.c_double * 3)(0.229, 0.224, 0.225)
        # normalised channels-last copy of the image (float32, or bfloat16 for the bf16 kernel): per call, so calls
        # on different streams (DRN.batch_predict(streams > 1)) never share the context-wide workspace
        scratch = torch.empty((B, H, W, 3), dtype=dtype, device=x.device)
        if split and dtype == torch.float32:
            am = torch.empty(1, dtype=torch.int32, device=x.device)      # the largest value stored: layer 2's scale
            if want_layer0:
                # DRN-C: layer0's output as well (the residual of layer1's BasicBlock)
                out0 = torch.empty((B, 16, H, W), dtype=dtype, device=x.device, memory_format=torch.channels_last)
                check(self._lib.spa_drn_stem_c_amax(self._ctx, _ptr(x), B, H, W, _ptr(w0), _ptr(b0), _ptr(w1p), _ptr(b1),
                                                    mean, std, _ptr(out), _ptr(out0), _ptr(scratch), _ptr(am), self._s()))
                out._spa_amax = am
                return out, out0
            check(self._lib.spa_drn_stem_d_amax(self._ctx, _ptr(x), B, H, W, _ptr(w0), _ptr(b0), _ptr(w1p), _ptr(b1),
                                                mean, std, _ptr(out), _ptr(scratch), _ptr(am), self._s()))
            out._spa_amax = am
            return out
        if want_layer0:
            # DRN-C in the bf16 network: conv1's output as well (the residual of layer1's BasicBlock)
            assert dtype == torch.bfloat16, 'want_layer0: the split-plane float32 kernel or the bf16 kernel'
            out0 = torch.empty((B, 16, H, W), dtype=dtype, device=x.device, memory_format=torch.channels_last)
            check(self._lib.spa_drn_stem_c_bf16(self._ctx, _ptr(x), B, H, W, _ptr(w0), _ptr(b0), _ptr(w1p), _ptr(b1),
                                                mean, std, _ptr(out), _ptr(out0), _ptr(scratch), self._s()))
            return out, out0
        check(self._lib.spa_drn_stem_d(self._ctx, _ptr(x), B, H, W, _ptr(w0), _ptr(b0), _ptr(w1p), _ptr(b1),
                                       mean, std, _ptr(out), 0 if dtype == torch.float32 else 1, _ptr(scratch),
                                       self._s()))
        return out

    @staticmethod
    def layer2_planes(weight):
        """(32,16,3,3) -> (wp, inv_t) for drn_layer2_f16s: the MFMA A fragments of the two half-precision planes of t * w,
        [2 channel tiles][2 planes][5 steps][64 lanes][8]: lane = (channel n = lane & 15, k group g = lane >> 4), k group
        (step s, g) = tap 2s + g // 2 (tap 9: zero pad), input channels 8 (g & 1) .. + 7 (csrc/spa_stem.hip)."""
        assert tuple(weight.shape) == (32, 16, 3, 3)
        w = weight.detach().float()
        amax = float(w.abs().max().clamp_min(1e-30))
        t = 2.0 ** (14 - int(np.floor(np.log2(amax))))
        ws = (w.double() * t).float().reshape(32, 16, 9)                      # (n, c, tap)
        frag = torch.zeros((2, 5, 64, 8), dtype=torch.float32, device=w.device)
        for s_ in range(5):
            for g in range(4):
                tap = 2 * s_ + g // 2
                if tap > 8:
                    continue
                c0 = 8 * (g & 1)
                for ct in range(2):
                    frag[ct, s_, g * 16:(g + 1) * 16, :] = ws[ct * 16:(ct + 1) * 16, c0:c0 + 8, tap]
        h = frag.half()
        l = (frag - h.float()).half()
        return torch.stack([h, l], dim=1).contiguous(), float(1.0 / t)        # (2, 2, 5, 64, 8)

    def drn_layer2_f16s(self, x, wp, inv_t, bias, amax_in=None):
        """relu(conv3x3(x; 16 -> 32 channels, stride 2, padding 1) + bias) on the 16-bit matrix cores at float32 accuracy.
        Returns y with the device word of its largest value attached as `_spa_amax`."""
        B, C, H, W = x.shape
        assert C == 16 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert wp.dtype == torch.float16 and tuple(wp.shape) == (2, 2, 5, 64, 8) and wp.is_contiguous()
        assert bias.dtype == torch.float32 and bias.numel() == 32 and bias.is_contiguous()
        if amax_in is None:
            amax_in = self.amax(x)
        y = torch.empty((B, 32, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        am = torch.empty(1, dtype=torch.int32, device=x.device)
        check(self._lib.spa_drn_layer2_f16s(self._ctx, _ptr(x), B, H, W, _ptr(wp), ctypes.c_float(inv_t), _ptr(bias), _ptr(amax_in),
                                            _ptr(am), _ptr(y), self._s()))
        y._spa_amax = am
        return y

    def drn_layer2_f32(self, x, w9, bias):
        """relu(conv3x3(x; 16 -> 32 channels, stride 2, padding 1) + bias) in plain float32 (the strict float32 network);
        w9 (9,16,32) float32 = (tap, input channel, output channel)"""
        B, C, H, W = x.shape
        assert C == 16 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert w9.dtype == torch.float32 and tuple(w9.shape) == (9, 16, 32) and w9.is_contiguous()
        assert bias.dtype == torch.float32 and bias.numel() == 32 and bias.is_contiguous()
        y = torch.empty((B, 32, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        check(self._lib.spa_drn_layer2_f32(self._ctx, _ptr(x), B, H, W, _ptr(w9), _ptr(bias), _ptr(y), self._s()))
        return y

    def conv3x3_s2_f32(self, x, wt, bias, csplit, relu=True):
        """conv3x3_s2_f16s with float32 matrix instructions: wt (Cout,9,Cin) float32 (the projection's rows hold its weights at
        tap 4).  Returns (y, y2 or None)."""
        B, Cin, Hi, Wi = x.shape
        Cout = wt.shape[0]
        assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert wt.dtype == torch.float32 and wt.is_contiguous() and tuple(wt.shape) == (Cout, 9, Cin)
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == Cout
        Ho, Wo = (Hi + 1) // 2, (Wi + 1) // 2
        y = torch.empty((B, csplit, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        y2 = torch.empty((B, Cout - csplit, Ho, Wo), dtype=torch.float32, device=x.device,
                         memory_format=torch.channels_last) if csplit < Cout else None
        check(self._lib.spa_conv3x3_s2_f32(self._ctx, _ptr(x), B, Hi, Wi, Cin, _ptr(wt), Cout, int(csplit), _ptr(bias),
                                           1 if relu else 0, _ptr(y), _ptr(y2), self._s()))
        return y, y2

    @staticmethod
    def small_planes(weight, proj_weight=None):
        """(Cout,Cin,3,3) [+ the block's 1x1 projection (Cp,Cin,1,1)] -> (wp, inv_t) for conv_small_f16s: the MFMA A fragments of
        the two half-precision planes of t * w.  Main tiles [Cout/16][2 planes][steps][64 lanes][8]: lane = (channel n = lane & 15
        of the tile, k group g = lane >> 4); Cin 16: 5 steps, k group (s, g) = tap 2s + g // 2 (tap 9: zero pad), input channels
        8 (g & 1) .. + 7; Cin 32: 9 steps, k group (s, g) = tap s, channels 8 g .. + 7.  Projection tiles [Cp/16][2][64][8]: the
        fragment of the step holding the centre tap (csrc/spa_convs.hip)."""
        Cout, Cin = int(weight.shape[0]), int(weight.shape[1])
        assert tuple(weight.shape[2:]) == (3, 3) and Cin in (16, 32) and Cout % 16 == 0
        w = weight.detach().float()
        amax = float(w.abs().max())
        if proj_weight is not None:
            assert tuple(proj_weight.shape[1:]) == (Cin, 1, 1) and proj_weight.shape[0] % 16 == 0 and Cin == 16
            amax = max(amax, float(proj_weight.detach().float().abs().max()))
        t = 2.0 ** (14 - int(np.floor(np.log2(max(amax, 1e-30)))))
        ws = (w.double() * t).float().reshape(Cout, Cin, 9)
        steps = 5 if Cin == 16 else 9
        frag = torch.zeros((Cout // 16, steps, 64, 8), dtype=torch.float32, device=w.device)
        for s_ in range(steps):
            for g in range(4):
                tap, c0 = (2 * s_ + g // 2, 8 * (g & 1)) if Cin == 16 else (s_, 8 * g)
                if tap > 8:
                    continue
                for ct in range(Cout // 16):
                    frag[ct, s_, g * 16:(g + 1) * 16, :] = ws[ct * 16:(ct + 1) * 16, c0:c0 + 8, tap]
        h = frag.half()
        parts = [torch.stack([h, (frag - h.float()).half()], dim=1).reshape(-1)]          # (tiles, 2, steps, 64, 8)
        if proj_weight is not None:
            Cp = int(proj_weight.shape[0])
            ps = (proj_weight.detach().double() * t).float().reshape(Cp, Cin)
            pf = torch.zeros((Cp // 16, 64, 8), dtype=torch.float32, device=w.device)
            for g in range(2):                                                            # tap 4 = step 2, k groups 0 and 1
                for pt in range(Cp // 16):
                    pf[pt, g * 16:(g + 1) * 16, :] = ps[pt * 16:(pt + 1) * 16, 8 * g:8 * g + 8]
            ph = pf.half()
            parts.append(torch.stack([ph, (pf - ph.float()).half()], dim=1).reshape(-1))  # (tiles, 2, 64, 8)
        return torch.cat(parts).contiguous(), float(1.0 / t)

    def conv_small_f16s(self, x, wp, inv_t, bias, cout, stride=1, n_proj=0, residual=None, relu=True, amax_in=None):
        """relu?(conv3x3(x; Cin 16 | 32 -> cout 16 | 32, stride 1 | 2, padding 1) + bias [+ residual]) on the 16-bit matrix cores
        at float32 accuracy, optionally with the block's 1x1 stride-2 projection as a second output (n_proj = 32).
        -> (y, y2 or None); y carries the device word of its largest value as `_spa_amax`."""
        B, Cin, H, W = x.shape
        assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert wp.dtype == torch.float16 and wp.is_contiguous() and bias.dtype == torch.float32 and bias.numel() == cout + n_proj
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        if amax_in is None:
            amax_in = self.amax(x)
        y = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        y2 = torch.empty((B, n_proj, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last) if n_proj else None
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.shape == y.shape and residual.is_contiguous(memory_format=torch.channels_last)
        am = torch.empty(1, dtype=torch.int32, device=x.device)
        check(self._lib.spa_conv_small_f16s(self._ctx, _ptr(x), B, H, W, Cin, _ptr(wp), ctypes.c_float(inv_t), int(cout), int(stride),
                                            int(n_proj), _ptr(bias), _ptr(residual), 1 if relu else 0, _ptr(amax_in), _ptr(am),
                                            _ptr(y), _ptr(y2), self._s()))
        y._spa_amax = am
        return y, y2

    def bias_act_(self, y, bias, residual=None, relu=True, track_amax=False):
        """In place y = relu?(y + bias [+ residual]) on a channels-last (B,C,H,W) activation.  track_amax (float32): the
        pass also records the largest magnitude it stores; the device word is attached to y as `_spa_amax`."""
        B, C, H, W = y.shape
        if track_amax and y.dtype == torch.float32:
            am = torch.empty(1, dtype=torch.int32, device=y.device)
            check(self._lib.spa_bias_act_amax(self._ctx, _ptr(y), B * H * W, C, _ptr(bias), _ptr(residual), 1 if relu else 0,
                                              _ptr(am), self._s()))
            y._spa_amax = am
            return y
        check(self._lib.spa_bias_act(self._ctx, _ptr(y), 0 if y.dtype == torch.float32 else 1, B * H * W, C,
                                     _ptr(bias), _ptr(residual), 1 if relu else 0, self._s()))
        return y

    def conv3x3_f32(self, x, wt, bias, residual=None, relu=True, dilation=1):
        """relu?(conv3x3(x; stride 1, padding = dilation) + bias [+ residual]) on the float32 matrix cores.
        x (B,Cin,H,W) float32 in channels-last storage, wt (Cout,9,Cin) float32, bias (Cout) float32."""
        B, Cin, H, W = x.shape
        Cout = wt.shape[0]
        assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        taps = wt.shape[1]
        assert wt.dtype == torch.float32 and wt.is_contiguous() and tuple(wt.shape) == (Cout, taps, Cin) and taps in (1, 9)
        assert bias.dtype == torch.float32 and bias.is_contiguous()
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.shape == y.shape and \
                residual.is_contiguous(memory_format=torch.channels_last)
        if taps == 1:           # (Cout, 1, Cin): the 1x1 projection
            check(self._lib.spa_conv1x1_f32(self._ctx, _ptr(x), B, H, W, Cin, _ptr(wt), Cout, _ptr(bias),
                                            _ptr(residual), 1 if relu else 0, _ptr(y), self._s()))
        else:
            check(self._lib.spa_conv3x3_f32(self._ctx, _ptr(x), B, H, W, Cin, _ptr(wt), Cout, _ptr(bias),
                                            _ptr(residual), 1 if relu else 0, int(dilation), _ptr(y), self._s()))
        return y

    @staticmethod
    def split_planes(wt):
        """(Cout,taps,Cin) float32 -> (wt2, inv_t) for conv3x3_f16s: wt2 (Cout,taps,Cin/32,2,32) float16 = the planes
        h = rn(t w), l = rn(t w - h) of the weights scaled by the power of two t that brings the largest magnitude into
        [2^14, 2^15); inv_t = 1 / t."""
        Cout, taps, Cin = wt.shape
        amax = float(wt.abs().max().clamp_min(1e-30))
        t = 2.0 ** (14 - int(np.floor(np.log2(amax))))
        ws = (wt.double() * t).float()                                        # exact
        h = ws.half()
        l = (ws - h.float()).half()
        wt2 = torch.stack([h.view(Cout, taps, Cin // 32, 32), l.view(Cout, taps, Cin // 32, 32)], dim=3).contiguous()
        return wt2, float(1.0 / t)

    def conv3x3_f16s(self, x, wt2, inv_t, bias, residual=None, relu=True, dilation=1, amax_in=None, track_amax=True):
        """conv3x3_f32 (3x3 or, with one tap, the 1x1 projection) on the 16-bit matrix cores at float32 accuracy: weights as
        two half-precision planes (split_planes), pixels split in the kernel.  Returns (y, amax of y or None)."""
        B, Cin, H, W = x.shape
        Cout, taps = wt2.shape[0], wt2.shape[1]
        assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert wt2.dtype == torch.float16 and wt2.is_contiguous() and tuple(wt2.shape) == (Cout, taps, Cin // 32, 2, 32) and taps in (1, 9)
        assert bias.dtype == torch.float32 and bias.is_contiguous()
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.shape == y.shape and \
                residual.is_contiguous(memory_format=torch.channels_last)
        if amax_in is None:
            amax_in = self.amax(x)
        amax_out = torch.empty(1, dtype=torch.int32, device=x.device) if track_amax else None
        if taps == 1:
            # (a 1x1 convolution does not see rows: the image goes in as ONE row of H * W pixels, so the kernel's 128- / 256-pixel
            # tiles are full whatever the width — the same sums per pixel, hence the same bits)
            H, W = 1, H * W
            check(self._lib.spa_conv1x1_f16s(self._ctx, _ptr(x), B, H, W, Cin, _ptr(wt2), ctypes.c_float(inv_t), Cout, _ptr(bias),
                                             _ptr(residual), 1 if relu else 0, _ptr(amax_in), _ptr(amax_out), _ptr(y), self._s()))
        else:
            check(self._lib.spa_conv3x3_f16s(self._ctx, _ptr(x), B, H, W, Cin, _ptr(wt2), ctypes.c_float(inv_t), Cout, _ptr(bias),
                                             _ptr(residual), 1 if relu else 0, int(dilation), _ptr(amax_in), _ptr(amax_out),
                                             _ptr(y), self._s()))
        return y, amax_out

    def conv3x3_s2_f16s(self, x, wt2, inv_t, bias, csplit, relu=True, amax_in=None):
        """3x3 stride-2 padding-1 convolution (+ the block's 1x1 stride-2 projection as output channels csplit..) on the
        16-bit matrix cores at float32 accuracy: one pass over x.  wt2 = split_planes of the (Cout,9,Cin) weights (the
        projection's rows hold its weights at tap 4).  Returns (y, y2 or None, amax of y)."""
        B, Cin, Hi, Wi = x.shape
        Cout = wt2.shape[0]
        assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert wt2.dtype == torch.float16 and wt2.is_contiguous() and tuple(wt2.shape) == (Cout, 9, Cin // 32, 2, 32)
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == Cout
        Ho, Wo = (Hi + 1) // 2, (Wi + 1) // 2
        y = torch.empty((B, csplit, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        y2 = torch.empty((B, Cout - csplit, Ho, Wo), dtype=torch.float32, device=x.device,
                         memory_format=torch.channels_last) if csplit < Cout else None
        if amax_in is None:
            amax_in = self.amax(x)
        amax_out = torch.empty(1, dtype=torch.int32, device=x.device)
        check(self._lib.spa_conv3x3_s2_f16s(self._ctx, _ptr(x), B, Hi, Wi, Cin, _ptr(wt2), ctypes.c_float(inv_t), Cout, int(csplit),
                                            _ptr(bias), 1 if relu else 0, _ptr(amax_in), _ptr(amax_out), _ptr(y), _ptr(y2), self._s()))
        return y, y2, amax_out

    @staticmethod
    def winograd_weights(weight, tile=2):
        """(Cout,Cin,3,3) -> (n*n,Cout,Cin) float32: G g G^T of F(tile x tile, 3x3), computed in float64, position-major.
        tile 2: points 0, 1, -1, inf (n = 4); tile 4: points 0, 1, -1, 1/2, -2, inf (n = 6; csrc/spa_wino.hip)."""
        if tile == 2:
            G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        else:
            G = [[1 / 2, 0, 0], [1 / 6, 1 / 6, 1 / 6], [1 / 6, -1 / 6, 1 / 6], [16 / 15, 8 / 15, 4 / 15],
                 [1 / 30, -1 / 15, 2 / 15], [0, 0, 1 / 2]]
        G = torch.tensor(G, dtype=torch.float64, device=weight.device)
        u = torch.einsum('ij,kcjl,ml->imkc', G, weight.detach().double(), G)
        return u.reshape(G.shape[0] ** 2, weight.shape[0], weight.shape[1]).float().contiguous()

    def conv3x3_wino_f32(self, x, u, bias, residual=None, relu=True, dilation=1):
        """relu?(conv3x3(x; stride 1, padding = dilation) + bias [+ residual]) by Winograd F(2x2,3x3) (u of 16
        positions) or F(4x4,3x3) (36 positions) on the float32 matrix cores.  x (B,Cin,H,W) float32 channels-last,
        u = winograd_weights(weight, tile), bias (Cout) float32."""
        B, Cin, H, W = x.shape
        npos, Cout = u.shape[0], u.shape[1]
        assert npos in (16, 36)
        assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert u.dtype == torch.float32 and u.is_contiguous() and tuple(u.shape) == (npos, Cout, Cin)
        assert bias.dtype == torch.float32 and bias.is_contiguous()
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.shape == y.shape and \
                residual.is_contiguous(memory_format=torch.channels_last)
        tiles, fn = (self._lib.spa_wino_tiles, self._lib.spa_conv3x3_wino_f32) if npos == 16 else \
            (self._lib.spa_wino4_tiles, self._lib.spa_conv3x3_wino4_f32)
        T = int(tiles(B, H, W, int(dilation)))
        v = torch.empty((npos, T, Cin), dtype=torch.float32, device=x.device)       # per call: stream safe
        m = torch.empty((npos, T, Cout), dtype=torch.float32, device=x.device)
        check(fn(self._ctx, _ptr(x), B, H, W, Cin, _ptr(u), Cout, _ptr(bias), _ptr(residual),
                 1 if relu else 0, int(dilation), _ptr(v), _ptr(m), _ptr(y), self._s()))
        return y

    @staticmethod
    def winograd_weights_split(weight):
        """(Cout,Cin,3,3) -> (u2, cs) for conv3x3_wino_f16s: u2 (36,Cout,Cin/32,2,32) float16 = the two half-precision
        planes h = rn(t U), l = rn(t U - h) of the F(4x4,3x3) weights U = G g G^T (float64 -> float32) scaled per position by
        the power of two t that brings the largest magnitude into [2^14, 2^15); cs = 36 float32 2^(p_i + p_j) / t_ij
        (csrc/spa_wino.hip)."""
        u = Engine.winograd_weights(weight, 4)                                # (36, Cout, Cin) float32
        Cout, Cin = u.shape[1], u.shape[2]
        amax = u.abs().amax(dim=(1, 2)).double().clamp_min(1e-30)
        t = torch.exp2(14.0 - torch.floor(torch.log2(amax)))                  # per position, exact powers of two
        us = (u.double() * t.view(36, 1, 1)).float()                          # exact
        h = us.half()
        l = (us - h.float()).half()
        u2 = torch.stack([h.view(36, Cout, Cin // 32, 32), l.view(36, Cout, Cin // 32, 32)], dim=3).contiguous()
        p = torch.tensor([4, 4, 4, 3, 3, 4], dtype=torch.float64, device=u.device)
        cs = (torch.exp2(p.view(6, 1) + p.view(1, 6)).reshape(36) / t).float().cpu().numpy().copy()
        return u2, cs

    def amax(self, x):
        """device word with the bit pattern of max |x| (the scale input of conv3x3_wino_f16s)"""
        assert x.dtype == torch.float32 and x.numel() % 4 == 0
        a = torch.empty(1, dtype=torch.int32, device=x.device)
        check(self._lib.spa_amax_f32(self._ctx, _ptr(x), x.numel(), _ptr(a), self._s()))
        return a

    def conv3x3_wino_f16s(self, x, u2, cs, bias, residual=None, relu=True, dilation=1, amax_in=None, track_amax=True, _keep=None):
        """conv3x3_wino_f32's F(4x4,3x3) with the GEMMs on the 16-bit matrix cores at float32 accuracy (two half-precision
        planes per operand, three products).  (u2, cs) = winograd_weights_split(weight).  Returns (y, amax of y or None);
        amax_in: the amax the producing call returned for x (computed here when None)."""
        B, Cin, H, W = x.shape
        Cout = u2.shape[1]
        assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        assert u2.dtype == torch.float16 and u2.is_contiguous() and tuple(u2.shape) == (36, Cout, Cin // 32, 2, 32)
        assert bias.dtype == torch.float32 and bias.is_contiguous() and cs.dtype == np.float32 and cs.shape == (36,)
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.shape == y.shape and \
                residual.is_contiguous(memory_format=torch.channels_last)
        if amax_in is None:
            amax_in = self.amax(x)
        amax_out = torch.empty(1, dtype=torch.int32, device=x.device) if track_amax else None
        T = int(self._lib.spa_wino4_tiles(B, H, W, int(dilation)))
        v = torch.empty((36, T, Cin), dtype=torch.float32, device=x.device)       # two float16 planes = 4 bytes per element
        m = torch.empty((36, T, Cout), dtype=torch.float32, device=x.device)
        if _keep is not None:                    # tools / tests: look at the transformed operands
            _keep['v'], _keep['m'] = v, m
        check(self._lib.spa_conv3x3_wino4_f16s(self._ctx, _ptr(x), B, H, W, Cin, _ptr(u2), cs.ctypes.data, Cout, _ptr(bias),
                                               _ptr(residual), 1 if relu else 0, int(dilation), _ptr(amax_in), _ptr(amax_out),
                                               _ptr(v), _ptr(m), _ptr(y), self._s()))
        return y, amax_out

    def conv3x3_bf16(self, x, wt, bias, residual=None, relu=True, dilation=1):
        """relu?(conv3x3(x; stride 1, padding = dilation) + bias [+ residual]) on the bf16 matrix cores.
        x (B,Cin,H,W) bfloat16 in channels-last storage, wt (Cout,9,Cin) bfloat16, bias (Cout) float32."""
        B, Cin, H, W = x.shape
        Cout = wt.shape[0]
        assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
        assert wt.dtype == torch.bfloat16 and wt.is_contiguous() and tuple(wt.shape) == (Cout, 9, Cin)
        assert bias.dtype == torch.float32 and bias.is_contiguous()
        y = torch.empty((B, Cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            assert residual.dtype == torch.bfloat16 and residual.shape == y.shape and \
                residual.is_contiguous(memory_format=torch.channels_last)
        check(self._lib.spa_conv3x3_bf16(self._ctx, _ptr(x), B, H, W, Cin, _ptr(wt), Cout, _ptr(bias),
                                         _ptr(residual), 1 if relu else 0, int(dilation), _ptr(y), self._s()))
        return y

    def conv_bf16_light(self, x, wt, bias, residual=None, relu=True, stride=1, dilation=1):
        """The light layers of the bf16 DRN (models/drn.py:134-151, 195-203) on libspalign's kernel: x (B,Cin,H,W) bf16
        channels-last, wt (Cout,taps,Cin) bf16 with taps 9 (3x3, padding = dilation) or 1 (1x1), stride 1 or 2."""
        B, Cin, H, W = x.shape
        Cout, taps = wt.shape[0], wt.shape[1]
        assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
        assert wt.dtype == torch.bfloat16 and wt.is_contiguous() and tuple(wt.shape) == (Cout, taps, Cin) and taps in (1, 9)
        assert bias.dtype == torch.float32 and bias.is_contiguous()
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        y = torch.empty((B, Cout, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            assert residual.dtype == torch.bfloat16 and residual.shape == y.shape and \
                residual.is_contiguous(memory_format=torch.channels_last)
        check(self._lib.spa_conv_bf16_light(self._ctx, _ptr(x), B, H, W, Cin, _ptr(wt), int(taps), int(stride), Cout, _ptr(bias),
                                            _ptr(residual), 1 if relu else 0, int(dilation), _ptr(y), self._s()))
        return y

    def resize_bicubic_u8(self, src, shape):
        """Decoded 8-bit images (B,H,W,C) uint8 -> (B,C,h,w) float32, bicubic exactly as Pillow's 8-bit
        Image.resize(..., BICUBIC) per channel (datasets/resize_image_dataset.py:31-34)."""
        src = _req(src, torch.uint8, 'src')
        B, H, W, C = src.shape
        h, w = int(shape[0]), int(shape[1])
        out = torch.empty((B, C, h, w), dtype=torch.float32, device=src.device)
        check(self._lib.spa_resize_bicubic_u8(self._ctx, _ptr(src), B, H, W, C, h, w, _ptr(out), self._s()))
        return out

    def resize_cvcubic_u8(self, src, shape):
        """Decoded 8-bit images (B,H,W,C) uint8 -> (B,C,h,w) float32 holding the bytes of OpenCV's 8-bit INTER_CUBIC resize
        (the cv2 branch of chainercv.transforms.resize on the uint8 image: what the reference environment ran; not pinned,
        see include/spalign.h)."""
        src = _req(src, torch.uint8, 'src')
        B, H, W, C = src.shape
        h, w = int(shape[0]), int(shape[1])
        out = torch.empty((B, C, h, w), dtype=torch.float32, device=src.device)
        check(self._lib.spa_resize_cvcubic_u8(self._ctx, _ptr(src), B, H, W, C, h, w, _ptr(out), self._s()))
        return out

    def resize_u8(self, src, shape, backend='pil'):
        return self.resize_cvcubic_u8(src, shape) if backend == 'cv2' else self.resize_bicubic_u8(src, shape)

    # ------------------------------------------------------------------ SLIC
    def rgb2lab(self, rgb, ratio=0.1):
        rgb = _req(rgb, torch.float32, 'rgb')
        B, C, H, W = rgb.shape
        assert C == 3
        lab = torch.empty_like(rgb)
        check(self._lib.spa_rgb2lab(self._ctx, _ptr(rgb), B, H, W, ratio, _ptr(lab), self._s()))
        return lab

    def slic_core(self, lab, n_segments, max_iter=10, want_centres=False):
        lab = _req(lab, torch.float32, 'lab')
        B, C, H, W = lab.shape
        assert C == 3
        labels = torch.empty((B, H, W), dtype=torch.int32, device=lab.device)
        centres = None
        if want_centres:
            nC = _lib.make_plan(H, W, n_segments).n_centroids
            centres = torch.empty((B, nC, 6), dtype=torch.float32, device=lab.device)
        check(self._lib.spa_slic_core(self._ctx, _ptr(lab), B, H, W, n_segments, max_iter,
                                      _ptr(labels), _ptr(centres), self._s()))
        return (labels, centres) if want_centres else labels

    def enforce_connectivity(self, labels, min_size, max_size):
        labels = _req(labels, torch.int32, 'labels')
        B, H, W = labels.shape
        out = torch.empty_like(labels)
        n_labels = torch.empty((B,), dtype=torch.int32, device=labels.device)
        check(self._lib.spa_enforce_connectivity(self._ctx, _ptr(labels), B, H, W, min_size, max_size,
                                                 _ptr(out), _ptr(n_labels), self._s()))
        return out, n_labels

    def slic(self, rgb, n_segments, compactness=10.0, max_iter=10):
        """slic(img, n_segments) for a batch: (B,3,H,W) f32 0..255 -> labels (B,H,W) i32, n_labels (B)."""
        rgb = _req(rgb, torch.float32, 'rgb')
        B, C, H, W = rgb.shape
        assert C == 3
        labels = torch.empty((B, H, W), dtype=torch.int32, device=rgb.device)
        n_labels = torch.empty((B,), dtype=torch.int32, device=rgb.device)
        check(self._lib.spa_slic(self._ctx, _ptr(rgb), B, H, W, n_segments, compactness, max_iter,
                                 _ptr(labels), _ptr(n_labels), self._s()))
        return labels, n_labels

    def slic_u8(self, rgb, n_segments, compactness=10.0, max_iter=10):
        """slic(uint8 image, n_segments) as superpixel_overlaps.py:303 calls it: scikit-image's float64 core.
        rgb (B,3,H,W) f32 holding the uint8 values -> labels (B,H,W) i32, n_labels (B)."""
        rgb = _req(rgb, torch.float32, 'rgb')
        B, C, H, W = rgb.shape
        assert C == 3
        labels = torch.empty((B, H, W), dtype=torch.int32, device=rgb.device)
        n_labels = torch.empty((B,), dtype=torch.int32, device=rgb.device)
        check(self._lib.spa_slic_u8(self._ctx, _ptr(rgb), B, H, W, n_segments, float(compactness), max_iter,
                                    _ptr(labels), _ptr(n_labels), self._s()))
        return labels, n_labels

    def rgb2lab_u8_f64(self, rgb, compactness=10.0):
        """rgb2lab(img_as_float(uint8 image)) * (1/compactness), float64 planar (B,3,H,W)."""
        rgb = _req(rgb, torch.float32, 'rgb')
        B, C, H, W = rgb.shape
        lab = torch.empty((B, 3, H, W), dtype=torch.float64, device=rgb.device)
        check(self._lib.spa_rgb2lab_u8_f64(self._ctx, _ptr(rgb), B, H, W, 1.0 / float(compactness), _ptr(lab), self._s()))
        return lab

    def slic_core_f64(self, lab, n_segments, max_iter=10, want_centres=False):
        """_slic_cython[double] on a scaled float64 Lab image (B,3,H,W) -> labels (B,H,W) i32 [, centres (B,nC,6)]."""
        lab = _req(lab, torch.float64, 'lab')
        B, C, H, W = lab.shape
        labels = torch.empty((B, H, W), dtype=torch.int32, device=lab.device)
        cen = None
        if want_centres:
            cen = torch.empty((B, _lib.make_plan(H, W, n_segments).n_centroids, 6), dtype=torch.float64, device=lab.device)
        check(self._lib.spa_slic_core_f64(self._ctx, _ptr(lab), B, H, W, n_segments, max_iter, _ptr(labels),
                                          _ptr(cen), self._s()))
        return (labels, cen) if want_centres else labels

    def felzenszwalb(self, rgb, scale=300.0, sigma=0.8, min_size=20, uint8_image=False):
        """felzenszwalb(img/255, scale, sigma, min_size) for a batch -> labels (B,H,W) i32, n_labels (B).
        uint8_image: the reference passed a uint8 image, so /255. happens in float64
        (superpixel_overlaps.py:294-300) instead of float32 (batch_spalign_kmeans.py:301-307)."""
        rgb = _req(rgb, torch.float32, 'rgb')
        B, C, H, W = rgb.shape
        assert C == 3
        labels = torch.empty((B, H, W), dtype=torch.int32, device=rgb.device)
        n_labels = torch.empty((B,), dtype=torch.int32, device=rgb.device)
        fn = self._lib.spa_felzenszwalb_u8 if uint8_image else self._lib.spa_felzenszwalb
        check(fn(self._ctx, _ptr(rgb), B, H, W, scale, sigma, min_size, _ptr(labels), _ptr(n_labels), self._s()))
        return labels, n_labels

    def overlap_refine(self, labels, road, max_labels, threshold):
        """superpixel_overlaps.py:353-361 -> refined (B,H,W) u8 (1 = road)."""
        labels = _req(labels, torch.int32, 'labels')
        road = _req(road, torch.uint8, 'road')
        B = labels.shape[0]
        npix = labels[0].numel()
        assert road.shape == labels.shape
        out = torch.empty(labels.shape, dtype=torch.uint8, device=labels.device)
        check(self._lib.spa_overlap_refine(self._ctx, _ptr(labels), _ptr(road), B, npix, int(max_labels),
                                           float(threshold), _ptr(out), self._s()))
        return out

    # ------------------------------------------------------------------ descriptors
    def segment_offsets(self, n_labels):
        n_labels = _req(n_labels, torch.int32, 'n_labels')
        B = n_labels.numel()
        off = torch.empty((B + 1,), dtype=torch.int32, device=n_labels.device)
        check(self._lib.spa_segment_offsets(self._ctx, _ptr(n_labels), B, _ptr(off), self._s()))
        return off

    def segment_stats(self, labels, offsets, ncap, prior_params=None, want_centroid=True):
        """-> count (ncap) i32, centroid (ncap,2) f64 or None, prior (ncap) f64 or None."""
        labels = _req(labels, torch.int32, 'labels')
        offsets = _req(offsets, torch.int32, 'offsets')
        B, H, W = labels.shape
        dev = labels.device
        count = torch.empty((ncap,), dtype=torch.int32, device=dev)
        centroid = torch.zeros((ncap, 2), dtype=torch.float64, device=dev) if want_centroid else None
        prior = torch.zeros((ncap,), dtype=torch.float64, device=dev) if prior_params else None
        yp, xp, ys, xs = prior_params if prior_params else (0.75, 0.5, 0.1, 0.1)
        check(self._lib.spa_segment_stats(self._ctx, _ptr(labels), B, H, W, _ptr(offsets), ncap,
                                          yp, xp, ys, xs, _ptr(count), _ptr(centroid), _ptr(prior),
                                          self._s()))
        return count, centroid, prior

    # device-resident CPython `random` stream (random.seed(1111) of the reference's module scope)
    def pyrandom_seed(self, seed=1111):
        check(self._lib.spa_pyrandom_dev_seed(self._ctx, int(seed), self._s()))

    def pyrandom_generate(self, want):
        """Make `want` outputs of the stream available (asynchronous on the current stream)."""
        check(self._lib.spa_pyrandom_dev_generate(self._ctx, int(want), self._s()))

    def anchor_ranks(self, count, n_ptr, ncap, n_anchors, total_pixels):
        """random.shuffle(pixel list)[:n_anchors] of every superpixel, on the device:
        -> ranks (ncap, n_anchors) int32 (raster rank of the chosen pixels), n_valid (ncap) int32."""
        count = _req(count, torch.int32, 'count')
        ranks = torch.empty((ncap, n_anchors), dtype=torch.int32, device=count.device)
        n_valid = torch.zeros((ncap,), dtype=torch.int32, device=count.device)
        check(self._lib.spa_anchor_ranks_dev(self._ctx, _ptr(count), _ptr(n_ptr), ncap, n_anchors, int(total_pixels),
                                             _ptr(ranks), _ptr(n_valid), self._s()))
        return ranks, n_valid

    def select_anchor_pixels(self, labels, offsets, ncap, ranks, n_valid):
        labels = _req(labels, torch.int32, 'labels')
        ranks = _req(ranks, torch.int32, 'ranks')
        n_valid = _req(n_valid, torch.int32, 'n_valid')
        B, H, W = labels.shape
        A = ranks.shape[1]
        anchors = torch.zeros((ncap, A, 2), dtype=torch.int32, device=labels.device)
        check(self._lib.spa_select_anchor_pixels(self._ctx, _ptr(labels), B, H, W, _ptr(offsets), ncap,
                                                 _ptr(ranks), _ptr(n_valid), A, _ptr(anchors),
                                                 self._s()))
        return anchors

    @staticmethod
    def fmap_desc(fmap):
        """fmap: (B, C, fh, fw) tensor in channels_last memory format (float32 or bfloat16)."""
        if fmap.dtype not in (torch.float32, torch.bfloat16):
            raise SpalignError('feature map dtype must be float32 or bfloat16')
        B, C, fh, fw = fmap.shape
        sb, sc, sy, sx = fmap.stride()
        return FmapDesc(C, fh, fw, sb, sc, sy, sx, 0 if fmap.dtype == torch.float32 else 1)

    @staticmethod
    def as_channels_last(fmap):
        """NCHW-shaped tensor stored NHWC; a no-op for the DRN module of this package."""
        if fmap.stride(1) == 1 and fmap.is_contiguous(memory_format=torch.channels_last):
            return fmap
        return fmap.contiguous(memory_format=torch.channels_last)

    def _new_x(self, ncap, C, append_pos, x_dtype, dev):
        D = C + (2 if append_pos else 0)
        return torch.zeros((ncap, D), dtype=x_dtype, device=dev), D

    def pool_anchor(self, fmap, img_h, offsets, ncap, anchors, n_valid, n_neighbors=4,
                    centroid=None, append_pos=True):
        fmap = self.as_channels_last(fmap)
        d = self.fmap_desc(fmap)
        B = fmap.shape[0]
        x_dtype = torch.float64 if append_pos else torch.float32
        X, D = self._new_x(ncap, d.C, append_pos, x_dtype, fmap.device)
        A = anchors.shape[1]
        check(self._lib.spa_pool_anchor(self._ctx, _ptr(fmap), ctypes.byref(d), B, img_h,
                                        _ptr(offsets), ncap, _ptr(anchors), _ptr(n_valid), A,
                                        n_neighbors, _ptr(centroid), 1 if append_pos else 0, _ptr(X),
                                        1 if append_pos else 0, D, self._s()))
        return X

    def pool_mean(self, fmap, labels, offsets, ncap, count, sampling='nearest', centroid=None,
                  append_pos=True):
        fmap = self.as_channels_last(fmap)
        d = self.fmap_desc(fmap)
        B, H, W = labels.shape
        x_dtype = torch.float64 if append_pos else torch.float32
        X, D = self._new_x(ncap, d.C, append_pos, x_dtype, fmap.device)
        check(self._lib.spa_pool_mean(self._ctx, _ptr(fmap), ctypes.byref(d), _ptr(labels), B, H, W,
                                      _ptr(offsets), ncap, _ptr(count),
                                      {'nearest': 0, 'bilinear': 1}[sampling], _ptr(centroid),
                                      1 if append_pos else 0, _ptr(X), 1 if append_pos else 0, D,
                                      self._s()))
        return X

    # ------------------------------------------------------------------ k-means + paint
    def kmeans(self, X, w, n_ptr, k, max_iter=1000, init_other=None, gate=None):
        """-> assign (ncap) i32, info (4) i32 {iterations, status, N, -} (both on the device).
        gate: device int32 word — the launch runs only if it holds a positive number (outputs stay zero otherwise)."""
        if X.dtype not in (torch.float32, torch.float64) or not X.is_contiguous():
            raise SpalignError('X must be contiguous float32/float64')
        w = _req(w, torch.float64, 'weights')
        ncap, D = X.shape
        assign = torch.zeros((ncap,), dtype=torch.int32, device=X.device)
        info = torch.zeros((4,), dtype=torch.int32, device=X.device)
        check(self._lib.spa_kmeans_weighted_gated(self._ctx, _ptr(X), 0 if X.dtype == torch.float32 else 1,
                                                  D, D, _ptr(w), _ptr(n_ptr), ncap, k, max_iter,
                                                  _ptr(init_other), _ptr(gate), _ptr(assign), _ptr(info), self._s()))
        return assign, info

    def retry_update(self, assign, offsets, B, counters, gate=None):
        """Retry bookkeeping of one k-means run on the device (spa_kmeans_retry_update): counters (2) int32 = [pending, made]."""
        check(self._lib.spa_kmeans_retry_update(self._ctx, _ptr(assign), _ptr(offsets), int(B), _ptr(gate), _ptr(counters[0:1]),
                                                _ptr(counters[1:2]), None, self._s()))

    NP_INIT_MAX = 65536          # csrc/spa_nprng.hip: the index vector lives in LDS, one byte per point

    def np_kmeans_init(self, state, w, n_ptr, k, gate=None):
        """The k > 2 initial assignment (batch_spalign_kmeans.py:141-149) drawn on the device from numpy's stream.
        state: (628,) int32 device tensor holding the generator (NpRandom.state() uploaded once; advanced in place);
        -> init_other (ncap) int64 for kmeans().  gate as in kmeans(): nothing is drawn unless it is positive."""
        w = _req(w, torch.float64, 'weights')
        ncap = w.shape[0]
        assert state.dtype == torch.int32 and state.numel() == 628 and state.is_contiguous() and ncap <= self.NP_INIT_MAX
        init = torch.zeros((ncap,), dtype=torch.int64, device=w.device)
        check(self._lib.spa_np_kmeans_init_dev(self._ctx, _ptr(state), _ptr(w), _ptr(n_ptr), ncap, k, _ptr(gate), _ptr(init),
                                               None, self._s()))
        return init

    def paint(self, labels, assign, offsets):
        labels = _req(labels, torch.int32, 'labels')
        B, H, W = labels.shape
        cluster = torch.empty((B, H, W), dtype=torch.uint8, device=labels.device)
        road = torch.empty((B, H, W), dtype=torch.uint8, device=labels.device)
        check(self._lib.spa_paint(self._ctx, _ptr(labels), _ptr(assign), _ptr(offsets), B, H, W,
                                  _ptr(cluster), _ptr(road), self._s()))
        return cluster, road

    def confusion(self, road, gt):
        """road (B,H,W) u8, gt (B,H,W) i32 in {-1,0,1} -> (B,4) i64 {TN, FP, FN, TP}."""
        road = _req(road, torch.uint8, 'road')
        gt = _req(gt, torch.int32, 'gt')
        B = road.shape[0]
        out = torch.zeros((B, 4), dtype=torch.int64, device=road.device)
        check(self._lib.spa_confusion(self._ctx, _ptr(road), _ptr(gt), B, road[0].numel(), _ptr(out),
                                      self._s()))
        return out


_DEFAULT = None


def default_engine():
    """Process-wide Engine (one spa_ctx per process/GPU)."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Engine()
    return _DEFAULT


class PyRandom(object):
    """CPython `random` stream (random.seed(1111); random.shuffle), host side."""

    def __init__(self, seed=1111):
        self._lib = _lib.lib()
        h = ctypes.c_void_p()
        check(self._lib.spa_pyrandom_create(seed, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        try:
            self._lib.spa_pyrandom_destroy(self._h)
        except Exception:
            pass

    def shuffle_select(self, counts, n_anchors):
        """counts: int32 numpy (N,) -> ranks (N, n_anchors) int32, n_valid (N,) int32."""
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        N = counts.size
        ranks = np.zeros((N, n_anchors), np.int32)
        n_valid = np.zeros((N,), np.int32)
        check(self._lib.spa_pyrandom_shuffle_select_host(
            self._h, counts.ctypes.data_as(ctypes.c_void_p), N, n_anchors,
            ranks.ctypes.data_as(ctypes.c_void_p), n_valid.ctypes.data_as(ctypes.c_void_p)))
        return ranks, n_valid


class NpRandom(object):
    """numpy legacy global RandomState stream (np.random.seed(1111); np.random.shuffle)."""

    def __init__(self, seed=1111):
        self._lib = _lib.lib()
        h = ctypes.c_void_p()
        check(self._lib.spa_nprandom_create(seed, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        try:
            self._lib.spa_nprandom_destroy(self._h)
        except Exception:
            pass

    def shuffle(self, a):
        assert a.dtype == np.int64 and a.flags.c_contiguous
        check(self._lib.spa_nprandom_shuffle_host(self._h, a.ctypes.data_as(ctypes.c_void_p), a.size))
        return a

    def state(self):
        """numpy's rk_state as it stands: (628,) uint32 = 624 state words, the position in the current block, padding —
        what Engine.np_kmeans_init keeps in device memory."""
        out = np.zeros(628, np.uint32)
        check(self._lib.spa_nprandom_state(self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def set_state(self, state628):
        """continue from a state taken with state() (or downloaded from the device copy np_kmeans_init advances)"""
        a = np.ascontiguousarray(state628, dtype=np.uint32)
        assert a.size == 628
        check(self._lib.spa_nprandom_set_state(self._h, a.ctypes.data_as(ctypes.c_void_p)))
