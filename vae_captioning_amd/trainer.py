"""VGG16 engine + the training-step harness (counterpart of main.py:186-290's inner loop).

Data parallelism (new work, the reference is single-GPU): one process per GPU,
replicated parameters, the minibatch sharded by image; ONE RCCL all-reduce per step over
the single flat gradient buffer (caption-side grads | reduction scalars | VGG grads),
plus a 4-byte all-reduce of the non-PAD token count that the CE gradient scale needs
before backward (main.py:156-157 divides by the GLOBAL count).
"""
import os

import numpy as np
import torch

from . import abi, spec
from .abi import ptr as P
from .engine import TAIL, CaptionEngine, FlatStore, _stream, internal_caption_variables, _round


def imagenet_weights(weight_file):
    """{cnn/* variable name: float32 array} from a vgg16_weights.npz, by the reference's rule
    (utils/image_embeddings.py:240-246, quirk Q18): the first 30 ALPHABETICALLY SORTED arrays (conv1_1_W, conv1_1_b ...
    conv5_3_b, fc6_W, fc6_b, fc7_W, fc7_b) are assigned to `parameters` in creation order; fc8_* are skipped."""
    names = [n for n, _ in spec.vgg_variables()]
    out = {}
    with np.load(weight_file) as w:
        for i, k in enumerate(sorted(w.keys())):
            if i == 30:
                break
            out[names[i]] = np.ascontiguousarray(w[k], dtype=np.float32)
    return out


class VggEngine(object):
    """utils/image_embeddings.py:26-238 on device: 13 x (conv3x3 + bias + ReLU), 5 max-pools,
    fc1 / fc2 (+ dropout), forward and backward, plus cnn_optimizer (ops/optimizers.py:49-82)."""

    def __init__(self, p, device="cuda", lib=None, grad_backing=None, seed=0, rank=0):
        self.p = p
        self.lib = lib or abi.load()
        self.dev = device
        self.store = FlatStore(spec.vgg_variables(), device, tail=0, grad_backing=grad_backing)
        self.buf = {}
        self.ws = None
        self.ws_bytes = 0
        self.tail_ws = {}
        self.train = bool(p.fine_tune) and p.mode == "training"
        self.keep = float(p.cnn_dropout) if self.train else 1.0  # main.py:67-73
        self.wd = float(p.weight_decay) if self.train else 0.0
        self.seed, self.rank = seed, rank
        self.inject = False
        self.timer = None
        self.precision = "f32"   # "bf16x3": fc1 / fc2 products and the weight gradients on the bf16 pipe (per call, like CaptionEngine.precision)
        # the weight gradient of layer l and the data-gradient chain (layer l, then l-1 ...) are independent:
        # wgrads run on a side stream so that the tail of one kernel (the last partial round of workgroups)
        # is filled by the other instead of idling the chip
        # Convolution dispatch ("which kernel runs which layer").  Activations between conv1_1 and pool5 are in the
        # C4 layout [B][C/4][H][W][4] (include/vaecap.h); the reference's NHWC order is restored at the fc1 boundary.
        #   conv1_1 (3 -> 64 channels, HBM-bound)      csrc/conv_first.hip
        #   every other 3x3 layer, forward / dgrad     Winograd F(4x4,3x3) (csrc/conv_wino4.hip) where vc_conv3x3_wino4_preferred, else F(2x2,3x3) (conv_wino.hip)
        #   weight gradient                            Winograd F(3x3,2x2) (csrc/conv_wino_wgrad.hip)
        #   VC_CONV_WINO=0, or shapes neither takes    the NHWC implicit-GEMM kernels of csrc/conv.hip behind layout conversions (slow; also the
        #                                              independent checker of tests/)
        self.use_conv1 = True
        self.use_wino = os.environ.get("VC_CONV_WINO", "1") != "0"
        # conv4_x / conv5_x forward + data gradient on a once-transformed input (csrc/conv_wino4.hip MODE 2; bit-identical to the fused
        # kernel).  Measured inside the cfg4 step (profiles/r06_wino4v_step.md): the main kernels shrink by 15-25 % but the 24 transform
        # launches per step sit on the chains; f32 25.85-25.93 ms against 25.81-25.88 fused (nothing), split-bf16 mode 21.39 against
        # 21.66 (its shorter weight-gradient stream leaves the chains critical).  Default: on in the split-bf16 mode only; VC_WINO4V=1 / 0 forces.
        self._wino4v_env = os.environ.get("VC_WINO4V")
        self.vws = {}   # conv chain -> workspace of the transformed input
        # Streams: 3 = two half-batch convolution chains + the weight gradients on a third stream (the tail of one launch is filled by
        # another stream's launch; the data-parallel gradient buckets are issued from the weight-gradient stream), 1 = serial
        nstreams = int(os.environ.get("VC_VGG_STREAMS", "3"))
        self._side = torch.cuda.Stream() if nstreams >= 2 else None
        self._side2 = torch.cuda.Stream() if nstreams >= 3 else None
        # Trainer.capture() sets this: a hipGraph capture of the backward pass's stream pattern (the weight-gradient stream waiting
        # for the chain streams layer after layer) crashes in hipStreamEndCapture (ROCm 7.2); forward-only captures are fine.  A
        # captured step therefore runs the VGG16 on the caller's stream alone.
        self.one_stream = False
        self.wino4 = set()
        self.part = torch.zeros(self.lib.vc_sumsq_blocks(), dtype=torch.float32, device=device)
        # sum(w^2) of the regulariser: the Adam update of step t leaves the per-workgroup sums of the NEW parameters, which are step t + 1's
        # w (0.16 ms per step saved: no second pass over 0.54 GB); invalid after any other write to the parameters
        # fc1 / fc2 (the tail of the flat buffer, 89 % of it) may be updated before the convolution layers: two launches, each
        # leaving its own per-block sums of w^2
        self.o_fc = self.store.offset("cnn/fc1/weights")
        self.w2_blocks = (self.lib.vc_adam_blocks(self.o_fc), self.lib.vc_adam_blocks(self.store.n - self.o_fc))
        self.w2_part = torch.zeros(sum(self.w2_blocks), dtype=torch.float32, device=device)
        self.w2_valid = False
        self.w2_cache = True     # False: always recompute sum(w^2) (a captured step: parameters restored behind the graph's back would go unnoticed)
        self._w2_version = -1

    @property
    def side(self):
        return None if self.one_stream else self._side

    @property
    def side2(self):
        return None if self.one_stream else self._side2

    def _b(self, name, shape, dtype=torch.float32):
        t = self.buf.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape:
            t = torch.zeros(shape, dtype=dtype, device=self.dev)
            self.buf[name] = t
            # the zero-fill is enqueued on the current stream: the side streams must not touch the new
            # buffer before it has run (first step only; buffers persist afterwards)
            cur = torch.cuda.current_stream()
            for s in (self.side, self.side2):
                if s is not None and s != cur:
                    s.wait_stream(cur)
        return t

    def _need_ws(self, nbytes):
        if nbytes > self.ws_bytes:
            self.ws = torch.empty(max(int(nbytes), 1 << 20) // 4 + 16, dtype=torch.float32, device=self.dev)
            self.ws_bytes = self.ws.numel() * 4

    def _chain_ws(self, i, nb):
        """Workspace of conv chain i (one per stream: chains run concurrently) for the K-split tail launches of the
        forward / data-gradient convolutions at nb images."""
        key = (i, nb)
        t = self.tail_ws.get(key)
        if t is None:
            lib, need, H, W = self.lib, 0, 224, 224
            for name, ci, co in spec.VGG_CONV:
                cie = 4 if ci == 3 else ci
                need = max(need, lib.vc_conv3x3_fwd_workspace_bytes(nb, H, W, cie, co), lib.vc_conv3x3_dgrad_workspace_bytes(nb, H, W, cie, co))
                if name in spec.VGG_POOL_AFTER:
                    H, W = H // 2, W // 2
            t = self.tail_ws[key] = torch.empty(max(need, 16) // 4 + 16, dtype=torch.float32, device=self.dev)
        return t

    def gemm(self, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, flags=0):
        self._need_ws(self.lib.vc_gemm_workspace_bytes(M, N, K))
        self.lib.vc_gemm_f32(_stream(), ta, tb, M, N, K, P(A), lda, P(B), ldb, P(C), ldc, P(bias), flags | (4 if self.precision == "bf16x3" else 0),
                             P(self.ws), self.ws_bytes)

    def _timed(self, tag, flops, fn):
        if self.timer is not None:
            self.timer.run(tag, flops, fn)
        else:
            fn()

    def _pack_weights(self, backward, H=224, W=224, nb=1):
        """Transformed (G g G^T) copies of the 3x3 kernels in the Winograd kernels' operand order (forward layout, and the flipped
        + transposed one of the data gradient when a backward pass follows).  Runs on the weight-gradient stream, which is
        idle during the forward pass: the forward copies first, in layer order, each followed by the event its layer's
        launches wait for (returned as {layer: event}; conv1_1 needs none), then the data-gradient copies, which only the
        backward pass waits for (self.packed_bwd)."""
        self.packed_bwd = None
        self.wino4 = set()   # layers on the F(4x4,3x3) kernels this step
        if not self.use_wino:
            return None
        lib, S = self.lib, self.store
        main = torch.cuda.current_stream()
        st = self.side2 if self.side2 is not None else (self.side if self.side is not None else main)
        if st != main:
            st.wait_stream(main)
        evs = {}
        self._pack_events = evs   # (kept alive for the step: a hipGraph capture holds their records)
        with torch.cuda.stream(st):
            sh = _stream()
            for dgrad in ((0, 1) if backward else (0,)):
                h, w_ = H, W
                for name, ci, co in spec.VGG_CONV:
                    if ci % 32 == 0:
                        w = S.param(spec.vgg_var_names(name)[0])
                        made = True
                        if bool(lib.vc_conv3x3_wino4_preferred(nb, h, w_, ci, co)):
                            # F(4x4,3x3) where it is the faster form for launches over nb images (every layer of a block between two pools has
                            # the same H x W and at least 64 channels, so a block stays in one family: the ReLU bits pass from layer to layer)
                            self.wino4.add(name)
                            lib.vc_conv3x3_wino4_pack_f32(sh, ci, co, P(w), dgrad, P(self._b(("vpt_" if dgrad else "vp_") + name, (36 * ci * co,))))
                        elif bool(lib.vc_conv3x3_wino_supported(1, h, w_, ci, co, dgrad)):
                            lib.vc_conv3x3_wino_pack_f32(sh, ci, co, P(w), dgrad, P(self._b(("vpt_" if dgrad else "vp_") + name, (16 * ci * co,))))
                        else:    # neither family takes the shape: csrc/conv.hip reads the HWIO kernel itself
                            made = False
                        if not dgrad and made:
                            evs[name] = torch.cuda.Event()
                            evs[name].record(torch.cuda.current_stream())
                    if name in spec.VGG_POOL_AFTER:
                        h, w_ = h // 2, w_ // 2
            if backward:
                self.packed_bwd = torch.cuda.Event()
                self.packed_bwd.record(torch.cuda.current_stream())
        return evs

    def _wino_ok(self, name, nb, H, W, ci, co, dgrad):
        ok = self.lib.vc_conv3x3_wino4_supported if name in self.wino4 else self.lib.vc_conv3x3_wino_supported
        return self.use_wino and (("vpt_" if dgrad else "vp_") + name) in self.buf and bool(ok(nb, H, W, ci, co, dgrad))

    @property
    def use_wino4v(self):
        return self._wino4v_env != "0" if self._wino4v_env is not None else self.precision == "bf16x3"

    def _wino(self, name, entry, geom=None, ch=0):
        """The Winograd entry `entry` ("fwd_f32", "dgrad_bits_f32", "mask_words" ...) of the family that holds layer `name` this
        step: vc_conv3x3_wino4_* (F(4x4,3x3)) or vc_conv3x3_wino_* (F(2x2,3x3)) -- same arguments in both.
        geom = (nb, H, W, Cin, Cout, dgrad) of a forward / data-gradient LAUNCH on conv chain `ch`: where the library prefers it
        (vc_conv3x3_wino4v_preferred: conv4_x, conv5_x) the F(4x4,3x3) call runs on a once-transformed input -- the same entry with the
        chain's transform workspace appended, bit-identical results, same mask bits (csrc/conv_wino4.hip MODE 2)."""
        lib = self.lib
        if geom is not None and name in self.wino4 and self.use_wino4v and bool(lib.vc_conv3x3_wino4v_preferred(*geom)):
            nb, H, W, ci, co, dgrad = geom
            need = lib.vc_conv3x3_wino4v_workspace_bytes(nb, H, W, co if dgrad else ci)
            v = self.vws.get(ch)
            if v is None or v.numel() * 4 < need:
                v = self.vws[ch] = torch.empty(need // 4, dtype=torch.float32, device=self.dev)
            fn = getattr(lib, "vc_conv3x3_wino4v_" + entry)
            return lambda *a: fn(*a, P(v), v.numel() * 4)
        return getattr(lib, ("vc_conv3x3_wino4_" if name in self.wino4 else "vc_conv3x3_wino_") + entry)

    def _wino_wgrad_ok(self, B, H, W, ci, co):
        return self.use_wino and ci % 64 == 0 and co % 64 == 0 and bool(self.lib.vc_conv3x3_wino_wgrad_supported(B, H, W, ci, co))

    def _bx_wgrad_ok(self, B, H, W, ci, co):
        """split-bf16 mode (this engine's precision): the direct weight gradient on the bf16 matrix pipe (csrc/conv_wgrad_bx.hip) replaces
        the f32 Winograd F(3x3,2x2) kernel (VC_WGRAD_BX=0: A/B runs)"""
        return (self.use_wino and ci % 64 == 0 and co % 64 == 0 and self.precision == "bf16x3" and os.environ.get("VC_WGRAD_BX", "1") != "0"
                and bool(self.lib.vc_conv3x3_bx_wgrad_supported(B, H, W, ci, co)))

    def colsum(self, x, rows, cols, out):
        self._need_ws(self.lib.vc_colsum_workspace_bytes(rows, cols))
        self.lib.vc_colsum_f32(_stream(), P(x), rows, cols, cols, P(out), 0, P(self.ws), self.ws_bytes)

    def load_params(self, named):
        missing = [n for n in self.store.names() if n not in named]
        if missing:  # tf.train.Saver.restore raises NotFoundError for the same situation
            raise KeyError("checkpoint lacks %d cnn/* variable(s), e.g. %s -- it was written without the VGG16 variables" % (len(missing), missing[0]))
        self.w2_valid = False
        for name in self.store.names():
            dst = self.store.param(name)
            if tuple(np.shape(named[name])) != tuple(dst.shape):
                raise ValueError("%s: checkpoint shape %s, model shape %s" % (name, tuple(np.shape(named[name])), tuple(dst.shape)))
            dst.copy_(torch.from_numpy(np.ascontiguousarray(named[name], dtype=np.float32)))

    def load_weights(self, weight_file):
        """utils/image_embeddings.py:240-246: first 30 alphabetically sorted npz arrays."""
        self.w2_valid = False
        for name, arr in imagenet_weights(weight_file).items():
            self.store.param(name).copy_(torch.from_numpy(arr))

    def state_dict(self):
        torch.cuda.synchronize()
        return {n: self.store.param(n).detach().cpu().numpy().copy() for n in self.store.names()}

    def grads_dict(self):
        torch.cuda.synchronize()
        return {n: self.store.grad(n).detach().cpu().numpy().copy() for n in self.store.names()}

    def set_masks(self, drop1, drop2):
        self.inject = True
        for k, a in (("drop1", drop1), ("drop2", drop2)):
            self._b(k, a.shape).copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)))

    # ------------------------------------------------------------------ forward
    def _to_nhwc(self, tag, t, nb, H, W, C):
        """NHWC copy of the C4 tensor t (fallback path: the kernels of csrc/conv.hip are NHWC)."""
        out = self._b("nhwc_" + tag, (nb, H, W, C))
        self.lib.vc_c4_to_nhwc_f32(_stream(), nb, H, W, C, P(t), P(out))
        return out

    def forward(self, images, step=None):
        """images [B, 224, 224, 3] float32 RGB 0..255 on device -> fc2 [B, 4096]."""
        lib, st, S = self.lib, _stream(), self.store
        B, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        self.B = B
        x = self._b("x0", (B, H, W, 4))   # NHWC4 == C4 with one channel quad
        if images.dtype == torch.uint8:
            lib.vc_vgg_preprocess_u8(st, P(images), B, H, W, P(x))
        else:
            lib.vc_vgg_preprocess_f32(st, P(images), B, H, W, P(x))
        c1 = bool(self.use_conv1 and self.use_wino and lib.vc_conv1_supported(B, H, W))  # conv1_1 through csrc/conv_first.hip (unpadded weights)
        w4 = self._b("w1_4", (3, 3, 4, 64))
        if not c1:
            lib.vc_pad_dim_f32(st, P(S.param("cnn/conv1_1/weights")), 9, 3, 4, 64, P(w4))
        packed = self._pack_weights(self.train, H, W, B // 2 if (self.side is not None and B % 2 == 0 and B >= 2) else B)
        self.acts = []  # (layer name, input tensor, H, W, Cin_eff, Cout, weights used)
        self.mask_geom = {}  # layer name -> (images per launch, launches): forward launches that left their ReLU mask as bits
        self.mask_family = {}  # layer name -> 4 / 2: the Winograd family whose data gradient can read those bits (its lane order)
        # The conv / pool chain of one image is independent of every other image: with two streams the
        # batch is pushed through as two half-batch chains so that the tail of each kernel (its last partial
        # round of workgroups) overlaps the other chain's kernels.  Halves are contiguous slices of the leading (image) dimension.
        main = torch.cuda.current_stream()
        side = self.side if (self.side is not None and B % 2 == 0 and B >= 2) else None
        halves = [(0, B // 2, main), (B // 2, B // 2, side)] if side is not None else [(0, B, main)]
        if side is not None:
            side.wait_stream(main)
        for name, ci, co in spec.VGG_CONV:
            wn, bn = spec.vgg_var_names(name)
            cie = 4 if ci == 3 else ci
            w = w4 if ci == 3 else S.param(wn)
            y = self._b("y_" + name, (B, co // 4, H, W, 4))
            pooled = name in spec.VGG_POOL_AFTER
            yp = self._b("p_" + name, (B, co // 4, H // 2, W // 2, 4)) if pooled else None
            pool_bits = None   # set when every chain's forward of this pooled layer left routing codes
            for ch, (b0, nb, strm) in enumerate(halves):
                tws = self._chain_ws(ch, nb)
                with torch.cuda.stream(strm):
                    sh = _stream()
                    if ci == 3 and c1:  # conv1_1: its own HBM-bound kernel, unpadded weights
                        if (self.train and "conv1_2" in self.wino4 and H % 16 == 0 and W % 16 == 0
                                and lib.vc_conv3x3_wino_single_launch_supported(nb, H, W, co, co)):
                            # conv1_2's F(4x4,3x3) data gradient takes its ReLU mask as bits from here instead of re-reading this activation
                            mk = self._b("mk_%s_%d" % (name, ch), (lib.vc_conv3x3_wino4_mask_words(nb, H, W, co),), dtype=torch.int32)
                            self.mask_geom[name], self.mask_family[name] = (nb, len(halves)), 4
                            self._timed("conv_fwd", 2.0 * nb * H * W * 9 * ci * co,
                                        lambda: lib.vc_conv1_fwd_mask_f32(sh, nb, H, W, P(x[b0:]), P(S.param(wn)), P(S.param(bn)), P(y[b0:]), P(mk)))
                        else:
                            self._timed("conv_fwd", 2.0 * nb * H * W * 9 * ci * co,
                                        lambda: lib.vc_conv1_fwd_f32(sh, nb, H, W, P(x[b0:]), P(S.param(wn)), P(S.param(bn)), P(y[b0:]), 1))
                        continue
                    fl = 2.0 * nb * H * W * 9 * ci * co
                    if packed is not None and name in packed:   # this layer's transformed weights of this step are ready
                        torch.cuda.current_stream().wait_event(packed[name])
                    if self._wino_ok(name, nb, H, W, cie, co, 0):   # Winograd (calls over 2 GiB are cut into image ranges inside the library)
                        if self.train and not pooled and lib.vc_conv3x3_wino_single_launch_supported(nb, H, W, cie, co):
                            # the next layer is a convolution on this output: leave (y > 0) as bits in the lane order of ITS data gradient
                            mk = self._b("mk_%s_%d" % (name, ch), (self._wino(name, "mask_words")(nb, H, W, co),), dtype=torch.int32)
                            self.mask_geom[name] = (nb, len(halves))   # the bits are per tile of THIS launch geometry
                            self.mask_family[name] = 4 if name in self.wino4 else 2
                            self._timed("conv_fwd", fl, lambda: self._wino(name, "fwd_mask_f32", (nb, H, W, cie, co, 0), ch)(
                                sh, nb, H, W, cie, co, P(x[b0:]), P(self.buf["vp_" + name]), P(S.param(bn)), P(y[b0:]), 1, P(mk)))
                        elif pooled and self.train:
                            # the 2x2 max-pool is register math in the epilogue; it also leaves MaxPoolGrad's routing codes (4 bits per pooled
                            # element), so the backward pass does not re-read the pre-pool activation
                            pb = self._b("pb_" + name, (lib.vc_conv3x3_wino_pool_words(B, H, W, co),), dtype=torch.int32)
                            pool_bits = pb
                            w0 = b0 * (H // 2) * (W // 2) * (co // 8)
                            self._timed("conv_fwd", fl, lambda: self._wino(name, "fwd_pool_f32", (nb, H, W, cie, co, 0), ch)(
                                sh, nb, H, W, cie, co, P(x[b0:]), P(self.buf["vp_" + name]), P(S.param(bn)), P(y[b0:]), P(yp[b0:]), P(pb[w0:])))
                        else:
                            self._timed("conv_fwd", fl, lambda: self._wino(name, "fwd_f32", (nb, H, W, cie, co, 0), ch)(
                                sh, nb, H, W, cie, co, P(x[b0:]), P(self.buf["vp_" + name]), P(S.param(bn)), P(y[b0:]), P(yp[b0:]) if pooled else None, 1))
                        continue
                    # NHWC implicit-GEMM kernels of csrc/conv.hip behind layout conversions: any shape (VC_CONV_WINO=0, odd image sizes)
                    xn = self._to_nhwc("x_%s_%d" % (name, ch), x[b0:], nb, H, W, cie)   # (buffers keyed by layer: a shared name would be re-allocated at every shape change)
                    yn = self._b("nhwc_y_%s_%d" % (name, ch), (nb, H, W, co))
                    self._timed("conv_fwd", fl, lambda: lib.vc_conv3x3_fwd_f32(
                        sh, nb, H, W, cie, co, P(xn), P(w), P(S.param(bn)), P(yn), 1, P(tws), tws.numel() * 4))
                    lib.vc_nhwc_to_c4_f32(sh, nb, H, W, co, P(yn), P(y[b0:]))
                    if pooled:   # (B * co / 4 planes of H x W four-channel pixels: the NHWC kernel on C4 data)
                        lib.vc_maxpool2x2_fwd_f32(sh, nb * (co // 4), H, W, 4, P(y[b0:]), P(yp[b0:]))
            self.acts.append((name, x, H, W, cie, co, w))
            x = y
            if pooled:
                self.acts.append(("P", x, H, W, co, co, pool_bits))
                x = yp
                H, W = H // 2, W // 2
        if side is not None:
            main.wait_stream(side)
        # pool5 in the reference's order: [B, 7, 7, 512] NHWC == [B, 25088] (image_embeddings.py:222); 6.4 MB at 64 images
        co = int(x.shape[1]) * 4
        self.pool5 = x
        flat = self._b("flat", (B, H, W, co))
        lib.vc_c4_to_nhwc_f32(st, B, H, W, co, P(x), P(flat))
        self.flat = flat
        F1 = H * W * co
        fc1 = self._b("fc1", (B, 4096))
        fc1_bytes = 4.0 * (F1 * 4096 + B * F1 + B * 4096)   # the weight matrix (411 MB) + both activations: the product is HBM-bound at B <= 64
        self._timed("hbm_fc1_gemm", fc1_bytes,
                    lambda: self.gemm(0, 0, B, 4096, F1, flat, F1, S.param("cnn/fc1/weights"), 4096, fc1, 4096, S.param("cnn/fc1/biases"), 1))
        fc2 = self._b("fc2", (B, 4096))
        if self.keep < 1:
            if not self.inject:
                for i, k in enumerate(("drop1", "drop2")):
                    m = self._b(k, (B, 4096))
                    lib.vc_philox_bernoulli_f32(st, P(m), m.numel(), self.keep, self.seed * 1000003 + self.rank, (8 + i) << 32, P(step))
            fc1d = self._b("fc1d", (B, 4096))
            lib.vc_dropout_f32(st, P(fc1), P(self.buf["drop1"]), self.keep, B * 4096, P(fc1d))
        else:
            fc1d = fc1
        self.fc1d = fc1d
        self.gemm(0, 0, B, 4096, 4096, fc1d, 4096, S.param("cnn/fc2/weights"), 4096, fc2, 4096, S.param("cnn/fc2/biases"), 1)
        if self.keep < 1:
            fc2d = self._b("fc2d", (B, 4096))
            lib.vc_dropout_f32(st, P(fc2), P(self.buf["drop2"]), self.keep, B * 4096, P(fc2d))
        else:
            fc2d = fc2
        return fc2d

    def reg_sumsq(self, out_ptr):
        """sum(w^2) over every cnn/* variable (main.py:69-74, Q9: biases included) -> device scalar."""
        lib, st = self.lib, _stream()
        # the partial sums the previous step's Adam update left are good only while nobody else has written the parameters: torch
        # bumps the flat buffer's version counter on every in-place write through it or a view of it (load_state_dict, p.copy_, ...),
        # the library's own kernels do not
        if self.w2_valid and self.w2_cache and self.store.p._version == self._w2_version:
            lib.vc_reduce_sum_f32(st, P(self.w2_part), self.w2_part.numel(), 1.0, out_ptr, 0)
            return
        lib.vc_sumsq_partial_f32(st, P(self.store.p), self.store.n, P(self.part))
        lib.vc_reduce_sum_f32(st, P(self.part), self.part.numel(), 1.0, out_ptr, 0)

    # ------------------------------------------------------------------ backward
    def backward(self, dfc2, after_fc=None, after_layer=None):
        """after_fc: optional callback invoked as soon as the fc1 / fc2 gradients (89 % of the VGG
        gradient bytes) are final, so their all-reduce can overlap the convolution backward.
        after_layer: optional (conv layer name, callback): invoked on the weight-gradient stream right after
        that layer's weight gradient has been enqueued (every gradient of that layer and of the layers
        above it is then final in stream order)."""
        lib, st, S = self.lib, _stream(), self.store
        B = self.B
        m1 = P(self.buf["drop1"]) if self.keep < 1 else None
        m2 = P(self.buf["drop2"]) if self.keep < 1 else None
        d2 = self._b("d_fc2", (B, 4096))
        lib.vc_relu_bwd_f32(st, P(dfc2), P(self.buf["fc2"]), m2, self.keep, B * 4096, P(d2))
        self.gemm(1, 0, 4096, 4096, B, self.fc1d, 4096, d2, 4096, S.grad("cnn/fc2/weights"), 4096)
        self.colsum(d2, B, 4096, S.grad("cnn/fc2/biases"))
        d1 = self._b("d_fc1", (B, 4096))
        self.gemm(0, 1, B, 4096, 4096, d2, 4096, S.param("cnn/fc2/weights"), 4096, d1, 4096)
        lib.vc_relu_bwd_f32(st, P(d1), P(self.buf["fc1"]), m1, self.keep, B * 4096, P(d1))
        F1 = self.flat.numel() // B
        fc1_bytes = 4.0 * (F1 * 4096 + B * F1 + B * 4096)
        self._timed("hbm_fc1_gemm_bwd", fc1_bytes, lambda: self.gemm(1, 0, F1, 4096, B, self.flat, F1, d1, 4096, S.grad("cnn/fc1/weights"), 4096))
        self.colsum(d1, B, 4096, S.grad("cnn/fc1/biases"))
        dn = self._b("d_pool5", tuple(self.flat.shape))
        self._timed("hbm_fc1_gemm_bwd", fc1_bytes, lambda: self.gemm(0, 1, B, F1, 4096, d1, 4096, S.param("cnn/fc1/weights"), 4096, dn, F1))
        d = self._b("d_pool5_c4", tuple(self.pool5.shape))   # back into the convolution layers' C4 layout
        lib.vc_nhwc_to_c4_f32(st, B, int(dn.shape[1]), int(dn.shape[2]), int(dn.shape[3]), P(dn), P(d))
        if after_fc is not None:
            after_fc()
        dw4 = self._b("dw1_4", (3, 3, 4, 64))
        self._need_ws(max(lib.vc_conv1_wgrad_workspace_bytes(),
                          max(max(lib.vc_conv3x3_wgrad_workspace_bytes(B, a[2], a[3], a[4], a[5]),
                                  lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, a[2], a[3], a[4], a[5]) if self._wino_wgrad_ok(B, a[2], a[3], a[4], a[5]) else 0,
                                  lib.vc_conv3x3_bx_wgrad_workspace_bytes(B, a[2], a[3], a[4], a[5]) if self._bx_wgrad_ok(B, a[2], a[3], a[4], a[5]) else 0)
                              for a in self.acts if a[0] != "P")))
        main = torch.cuda.current_stream()
        side, side2 = self.side, self.side2
        # streams: with 3, the data-gradient chain runs as two half-batch chains (main, side) and every
        # weight gradient (full batch) on side2; with 2, one full-batch chain (main) + weight gradients (side)
        split = side2 is not None and B % 2 == 0
        wst = side2 if split else side
        halves = [(0, B // 2, main), (B // 2, B // 2, side)] if split else [(0, B, main)]
        if self.packed_bwd is not None:   # the data-gradient weight copies made during the forward pass
            main.wait_event(self.packed_bwd)
        if split:
            side.wait_stream(main)
        for li in range(len(self.acts) - 1, -1, -1):
            name, x, H, W, ci, co, w = self.acts[li]
            if name == "P":
                dx = self._b("dx_%d" % li, (B, co // 4, H, W, 4))
                for b0, nb, strm in halves:
                    with torch.cuda.stream(strm):  # + ReluGrad of the conv that made x
                        if w is not None:   # routing codes from the pooled Winograd forward: no read of x
                            lib.vc_maxpool2x2_bwd_bits_f32(_stream(), nb, H, W, co, P(w[b0 * (H // 2) * (W // 2) * (co // 8):]), P(d[b0:]), P(dx[b0:]))
                        else:   # (planes of four-channel pixels: the NHWC kernel on C4 data)
                            lib.vc_maxpool2x2_bwd_f32(_stream(), nb * (co // 4), H, W, 4, P(x[b0:]), P(d[b0:]), P(dx[b0:]), 1)
                d = dx
                continue
            wn, bn = spec.vgg_var_names(name)
            cr = 3 if ci == 4 else ci  # algorithmic channel count (conv1_1 is zero-padded 3 -> 4)
            fl = 2.0 * B * H * W * 9 * cr * co

            def wgrad(x=x, d=d, wn=wn, bn=bn, ci=ci, co=co, H=H, W=W, fl=fl):
                sw = _stream()
                if ci == 4 and self.use_conv1 and self.use_wino and lib.vc_conv1_supported(B, H, W):
                    self._timed("conv_wgrad", fl, lambda: lib.vc_conv1_wgrad_f32(sw, B, H, W, P(x), P(d), P(S.grad(wn)), P(S.grad(bn)), 0, P(self.ws), self.ws_bytes))
                elif ci != 4 and self._bx_wgrad_ok(B, H, W, ci, co):   # split-bf16 mode: direct, K = the pixels, operands split in registers
                    self._timed("conv_wgrad", fl, lambda: lib.vc_conv3x3_bx_wgrad_f32(sw, B, H, W, ci, co, P(x), P(d), P(S.grad(wn)), P(S.grad(bn)), 0, P(self.ws), self.ws_bytes))
                elif ci != 4 and self._wino_wgrad_ok(B, H, W, ci, co):   # Winograd F(3x3,2x2): both operands transformed in registers, K = the 2x2 tiles
                    self._timed("conv_wgrad", fl, lambda: lib.vc_conv3x3_wino_wgrad_f32(sw, B, H, W, ci, co, P(x), P(d), P(S.grad(wn)), P(S.grad(bn)), 0, P(self.ws), self.ws_bytes))
                else:   # csrc/conv.hip on NHWC copies (conv1_1: zero-padded 4-channel weights)
                    xn, dn_ = self._to_nhwc("wx_" + wn, x, B, H, W, ci), self._to_nhwc("wd_" + wn, d, B, H, W, co)
                    self._timed("conv_wgrad", fl, lambda: lib.vc_conv3x3_wgrad_f32(sw, B, H, W, ci, co, P(xn), P(dn_), P(dw4 if ci == 4 else S.grad(wn)), P(S.grad(bn)), 0,
                                                                                   P(self.ws), self.ws_bytes))
                    if ci == 4:
                        lib.vc_pad_dim_f32(sw, P(dw4), 9, 4, 3, 64, P(S.grad(wn)))
            if wst is not None:
                wst.wait_stream(main)  # d (this layer's pre-activation gradient) is final on the chain stream(s)
                if split:
                    wst.wait_stream(side)
                with torch.cuda.stream(wst):
                    wgrad()
                    if after_layer is not None and after_layer[0] == name:
                        after_layer[1]()
            else:
                wgrad()
                if after_layer is not None and after_layer[0] == name:
                    after_layer[1]()
            if li > 0:
                prev_is_pool = self.acts[li - 1][0] == "P"
                dx = self._b("dx_%d" % li, (B, ci // 4, H, W, 4))
                for ch, (b0, nb, strm) in enumerate(halves):
                    tws = self._chain_ws(ch, nb)
                    with torch.cuda.stream(strm):
                        sh = _stream()
                        if (self._wino_ok(name, nb, H, W, ci, co, 1) and not prev_is_pool
                              and self.mask_geom.get(self.acts[li - 1][0]) == (nb, len(halves))
                              and self.mask_family.get(self.acts[li - 1][0]) == (4 if name in self.wino4 else 2)   # (bits are in their family's lane order)
                              and lib.vc_conv3x3_wino_single_launch_supported(nb, H, W, ci, co)):
                            # ReluGrad from the bits the previous layer's forward left (one 8-byte load per lane instead of sixteen 16-byte ones)
                            self._timed("conv_dgrad", fl * nb / B, lambda: self._wino(name, "dgrad_bits_f32", (nb, H, W, ci, co, 1), ch)(
                                sh, nb, H, W, ci, co, P(d[b0:]), P(self.buf["vpt_" + name]), P(self.buf["mk_%s_%d" % (self.acts[li - 1][0], ch)]), P(dx[b0:])))
                        elif self._wino_ok(name, nb, H, W, ci, co, 1):
                            self._timed("conv_dgrad", fl * nb / B, lambda: self._wino(name, "dgrad_f32", (nb, H, W, ci, co, 1), ch)(
                                sh, nb, H, W, ci, co, P(d[b0:]), P(self.buf["vpt_" + name]), None if prev_is_pool else P(x[b0:]), P(dx[b0:])))
                        else:   # csrc/conv.hip on NHWC copies
                            dn_ = self._to_nhwc("d_%s_%d" % (name, ch), d[b0:], nb, H, W, co)
                            xn = None if prev_is_pool else self._to_nhwc("x_%s_%d" % (name, ch), x[b0:], nb, H, W, ci)
                            dxn = self._b("nhwc_dx_%s_%d" % (name, ch), (nb, H, W, ci))
                            self._timed("conv_dgrad", fl * nb / B, lambda: lib.vc_conv3x3_dgrad_f32(
                                sh, nb, H, W, ci, co, P(dn_), P(w), P(xn), P(dxn), P(tws), tws.numel() * 4))
                            lib.vc_nhwc_to_c4_f32(sh, nb, H, W, ci, P(dxn), P(dx[b0:]))
                d = dx
        if split:
            main.wait_stream(side)
        if wst is not None:
            main.wait_stream(wst)

    def apply_gradients(self, scal, part=None):
        """cnn_optimizer: no clipping; Adam(cnn_lr, beta1=0.8) by default; the L2 regulariser's
        gradient wd*w is folded into the update.  part: None = every cnn/* variable; "fc" = fc1 + fc2 only (their gradients are
        final as soon as the fc backward has run: the caller may update them on another stream under the convolution backward);
        "conv" = the convolution layers only.  The update is elementwise, so the two halves equal the whole."""
        p, lib, st, S = self.p, self.lib, _stream(), self.store
        for which, lo, n, w2o in (("conv", 0, self.o_fc, 0), ("fc", self.o_fc, S.n - self.o_fc, self.w2_blocks[0])):
            if part is not None and part != which:
                continue
            sl = lambda t: t.data_ptr() + lo * 4
            if p.cnn_optimizer == "Adam" and self.wd:
                self._timed("hbm_adam", 28.0 * n, lambda: lib.vc_adam_sumsq_f32(
                    st, sl(S.p), sl(S.g), sl(S.slot("m")), sl(S.slot("v")), n, scal.data_ptr() + 12, None, 0.8, 0.999, 1e-8, self.wd,
                    self.w2_part.data_ptr() + w2o * 4))
            elif p.cnn_optimizer == "Adam":
                self._timed("hbm_adam", 28.0 * n, lambda: lib.vc_adam_f32(
                    st, sl(S.p), sl(S.g), sl(S.slot("m")), sl(S.slot("v")), n, scal.data_ptr() + 12, None, 0.8, 0.999, 1e-8, self.wd))
            elif p.cnn_optimizer == "SGD":
                lib.vc_sgd_f32(st, sl(S.p), sl(S.g), n, scal.data_ptr() + 16, None, self.wd)
            else:
                lib.vc_momentum_f32(st, sl(S.p), sl(S.g), sl(S.slot("a")), n, scal.data_ptr() + 16, None, 0.9, self.wd, None, 0)
        # (every block of both halves has written its sum by the time the next forward pass reads them, whatever the order)
        self.w2_valid = p.cnn_optimizer == "Adam" and bool(self.wd)
        self._w2_version = self.store.p._version


class _NoPending(object):
    """Handle of a muted collective (Trainer.mute_collectives)."""

    def wait(self):
        pass


class Trainer(object):
    """One training step = main.py:241-244's sess.run([kld, rec_loss, lower_bound, optimize,
    optimize_cnn, annealing])."""

    def __init__(self, p, vocab, device="cuda", lib=None, world=1, rank=0, group=None, seed=0, force_collectives=False, comm="auto",
                 wgrad_stream=True, precision=None):
        """wgrad_stream: weight gradients of the caption side, its clip + optimiser and fc1 / fc2's optimiser run on a second
        stream (under the LSTM recurrences and the VGG16 backward pass); False = everything in program order on one stream --
        the same kernels on the same data either way, so the results are bit-identical.
        comm: how the data-parallel collectives run.  "abi" = libvaecap's own RCCL entries (dp.AbiComm: vc_allreduce_sum_f32 ...);
        "torch" = torch.distributed on `group`; "auto" = "abi" when the collectives are on and the process group is RCCL-backed
        (backend "nccl") or there is no process group at all (one forced rank), "torch" otherwise (the gloo test path).
        An existing dp.AbiComm may be passed instead."""
        self.p, self.lib = p, (lib or abi.load())
        # precision: "f32" (default: the reference's tf.float32 arithmetic) or "bf16x3" (split-bf16 operands, three bf16 MFMAs, f32
        # accumulate: every dense product of the step, the LSTM recurrences and the VGG16 weight gradients -- an opt-in mode with ~1e-5
        # relative product error, reported on its own bench lines, never the default).  None = VC_PRECISION, else "f32".  The mode
        # belongs to THIS trainer: its engines pass it with every library call (ABI 4), no process-wide state is read or written.
        precision = precision or os.environ.get("VC_PRECISION") or "f32"
        if precision not in ("f32", "bf16x3"):
            raise ValueError("precision must be 'f32' or 'bf16x3', not %r" % (precision,))
        self.precision = precision
        self.collectives = world > 1 or force_collectives
        self.world, self.rank, self.group = world, rank, group
        self.dev = device
        n_cap = sum(_round(int(np.prod(s))) for _, s in internal_caption_variables(p, vocab)) + TAIL
        self.fine = bool(p.fine_tune)
        n_vgg = sum(_round(int(np.prod(s))) for _, s in spec.vgg_variables()) if self.fine else 0
        self.gall = torch.zeros(n_cap + n_vgg, dtype=torch.float32, device=device)  # THE all-reduce buffer
        self.cap = CaptionEngine(p, vocab, device, self.lib, grad_backing=self.gall[:n_cap], world=world, rank=rank, group=group, seed=seed,
                                 force_collectives=force_collectives)
        self.cap.set_precision(self.precision)
        self.cap.enable_wgrad_stream(bool(wgrad_stream) and os.environ.get("VC_WGRAD_STREAM", "1") != "0")   # (VC_WGRAD_STREAM=0: A/B runs)
        self.vgg = None
        if self.fine:
            self.vgg = VggEngine(p, device, self.lib, grad_backing=self.gall[n_cap:], seed=seed, rank=rank)
            self.vgg.precision = self.precision
            if self.vgg.wd:
                self.cap.reg_scale = self.vgg.wd / 2.0  # l2_regularizer(wd)(w) = wd * sum(w^2)/2
        self.images = None
        self.graph = None
        self.dp_stats = None  # list: per-bucket (index, bytes, event before wait, event after wait) when bench.py asks for it
        self.cnn_host = None  # {cnn/* name: array} written into checkpoints when VGG16 is not on device (state_dict)
        self.n_cap = n_cap
        # Data-parallel gradient exchange.  Logically ONE sum-all-reduce of `gall` per step; with VGG
        # fine-tuning it is issued as four asynchronous pieces in the order the gradients become final
        # (caption side | fc1+fc2, 478 MB | conv3_1..conv5_3, 58 MB | conv1_1..conv2_2, 1 MB) so that RCCL
        # overlaps the convolution backward and only the last megabyte is exposed
        # (VC_DP_BUCKETS=0 forces the single blocking call).
        self.buckets = os.environ.get("VC_DP_BUCKETS", "1") != "0"
        self.off_fc = n_cap + self.vgg.store.offset("cnn/fc1/weights") if self.vgg is not None else None
        self.off_c3 = n_cap + self.vgg.store.offset("cnn/conv3_1/weights") if self.vgg is not None else None
        self.comm = None
        self.trace = os.environ.get("VC_TRACE", "0") == "1" and bool(self.lib.vc_trace_available())
        self._open_range = False
        if self.collectives:
            self._setup_comm(comm)

    def _setup_comm(self, comm):
        import torch.distributed as dist
        from . import dp
        if isinstance(comm, dp.AbiComm):
            self.comm = comm
        else:
            if comm == "auto":
                env = os.environ.get("VC_DP_COMM", "")
                if env in ("abi", "torch"):
                    comm = env
                elif dist.is_available() and dist.is_initialized():
                    comm = "abi" if dist.get_backend(self.group) == "nccl" else "torch"
                else:
                    comm = "abi" if self.world == 1 else "torch"
            if comm == "abi":
                dev = torch.cuda.current_device()
                if self.world == 1:
                    self.comm = dp.AbiComm.single(self.lib, dev)
                else:
                    self.comm = self._agree_on_abi_comm(dev)
        if self.comm is not None:   # the engine's collective hooks go through the C ABI
            self.cap.reduce_fn, self.cap.gather_fn, self.cap.rscatter_fn = self.comm.all_reduce, self.comm.all_gather, self.comm.reduce_scatter

    def _agree_on_abi_comm(self, dev):
        """The libvaecap communicator for world > 1, or None with every rank on torch.distributed -- the ranks must end up on the
        SAME path whatever fails where, so each stage is agreed on through the process group before the next one starts:
          1. vc_comm_available() on every rank (an RCCL library can be bound);
          2. rank 0 creates the unique id and broadcasts it -- or the error text -- to the group (broadcast_object_list: the group's
             own transport, no private store, nobody waits on a key that never appears);
          3. ncclCommInitRank on every rank (RCCL refuses e.g. two ranks on one GPU: an error on all of them), then a last MIN.
        (A rank that dies INSIDE the collective init still hangs the others -- that is RCCL's contract, not something a handshake
        in front of it can repair.)"""
        import torch.distributed as dist
        from . import dp
        on_gpu = dist.get_backend(self.group) == "nccl"

        def all_ok(ok):
            flag = torch.tensor([1.0 if ok else 0.0], device="cuda" if on_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return float(flag.item()) == 1.0

        def give_up(why):
            if self.rank == 0:
                print("libvaecap communicator not available on every rank (%s): collectives through torch.distributed" % why, flush=True)
            return None
        if not all_ok(bool(self.lib.vc_comm_available())):
            return give_up("no RCCL library could be bound on some rank")
        msg = [None]
        if self.rank == 0:
            try:
                msg = [("id", dp.AbiComm.unique_id(self.lib))]
            except abi.VaecapError as e:
                msg = [("error", str(e))]
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast_object_list(msg, src=src, group=self.group)
        if msg[0][0] != "id":
            return give_up(msg[0][1])
        comm, err = None, None
        try:
            comm = dp.AbiComm(self.lib, self.world, self.rank, dev, msg[0][1])
        except abi.VaecapError as e:
            err = e
        if not all_ok(err is None):
            if comm is not None:
                comm.destroy()
            return give_up(err or "another rank failed")
        return comm

    def set_batch(self, batch, noise=None):
        extra = []
        if self.fine:
            img = np.asarray(batch["images"])
            # uint8 pixels (what the reference's HDF5 holds, preprocess.py:27-28) travel as bytes -- 9.6 MB instead of 38.5 MB per 64
            # images -- and are cast on the device (vc_vgg_preprocess_u8); anything else is fed as float32 like the placeholder's cast
            extra = [("images", img, torch.uint8)] if img.dtype == np.uint8 else [("images", np.asarray(img, np.float32), torch.float32)]
        self.cap.set_batch(batch, noise, extra=extra)  # one pinned staging buffer, one asynchronous copy
        if self.fine:
            self.images = self.cap.buf["images"]
            if noise is not None and "cnn_drop1" in noise:
                self.vgg.set_masks(noise["cnn_drop1"], noise["cnn_drop2"])

    def _range(self, name):
        """roctx range around a phase of the step (VC_TRACE=1; `rocprofv3 --marker-trace`); name None closes the open range."""
        if self.trace:
            if self._open_range:
                self.lib.vc_trace_pop()
            if name is not None:
                self.lib.vc_trace_push(name.encode())
            self._open_range = name is not None

    def _step(self):
        cap, vgg = self.cap, self.vgg
        feats = None
        if vgg is not None:
            self._range("vgg16_forward")
            feats = vgg.forward(self.images, cap.step)
            if vgg.wd:
                vgg.reg_sumsq(cap.red.data_ptr() + 12)
        self._range("caption_forward")
        cap.forward(feats)
        self._range("caption_backward")
        dfe = cap.backward(want_dfeatures=vgg is not None)
        # Weight gradients, the clip and the optimisers are off the gradient chain (cap.off_chain: the weight-gradient stream when
        # Trainer has enabled it, else this stream): the caption side's run under the VGG16 backward pass, fc1 / fc2's Adam under
        # the convolution layers' backward.  Collectives keep their order: caption bucket, fc, conv3_1.., conv1_1..
        with cap.off_chain():
            cap.pack_tail()
        fine = vgg is not None and vgg.train
        self._range("vgg16_backward+allreduce" if fine else "allreduce")
        if self.collectives and self.buckets and fine and self.reduce_async_fn is not None:
            from . import dp
            bk = dp.gradient_buckets(self.n_cap, self.gall.numel(), self.off_fc, self.off_c3)
            issue = lambda i: self.reduce_async_fn(self.gall[bk[i][0]:bk[i][1]])

            def wait(i, h):
                if self.dp_stats is not None:  # how long the waiting stream stalls on each bucket (bench.py reports it)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    h.wait()
                    e1.record()
                    self.dp_stats.append((i, (bk[i][1] - bk[i][0]) * 4, e0, e1))
                else:
                    h.wait()

            def fc_final():
                with cap.off_chain():
                    wait(1, issue(1))
                    vgg.apply_gradients(cap.scal, "fc")

            pending = []
            with cap.off_chain():
                wait(0, issue(0))
                cap.apply_gradients()
            vgg.backward(dfe, after_fc=fc_final, after_layer=("conv3_1", lambda: pending.append((2, issue(2)))))
            pending.append((3, issue(3)))
            for i, h in pending:
                wait(i, h)
            self._range("optimizers")
            vgg.apply_gradients(cap.scal, "conv")
        else:
            if self.collectives:
                if fine:
                    vgg.backward(dfe)
                cap.join_off_chain()
                self.all_reduce_grads()  # the single gradient all-reduce (RCCL over xGMI)
                self._range("optimizers")
                with cap.off_chain():
                    cap.apply_gradients()
                if fine:
                    vgg.apply_gradients(cap.scal)
            else:
                with cap.off_chain():
                    cap.apply_gradients()
                if fine:
                    def fc_final():
                        with cap.off_chain():
                            vgg.apply_gradients(cap.scal, "fc")
                    vgg.backward(dfe, after_fc=fc_final)
                    self._range("optimizers")
                    vgg.apply_gradients(cap.scal, "conv")
        cap.join_off_chain()
        self._range(None)

    @property
    def reduce_async_fn(self):
        if getattr(self.cap, "_fake_collectives", False):
            return None
        if getattr(self, "_muted", False):   # mute_collectives: same bucket / stream structure, nothing on the wire
            return lambda t: _NoPending()
        if self.comm is not None:
            return self.comm.all_reduce_async
        return lambda t: torch.distributed.all_reduce(t, group=self.group, async_op=True)

    def mute_collectives(self):
        """Replace every collective of the step by a no-op (timing only: bench.py measures the compute-only step to report the
        exposed communication time = step - compute-only step).  The step keeps its structure -- the four gradient pieces are still
        "issued" and "waited for" where the real ones are, so the optimisers overlap the VGG16 backward pass exactly as in the real
        step.  Returns the function that restores the collectives."""
        cap = self.cap
        saved = (cap.reduce_fn, cap.gather_fn, cap.rscatter_fn)
        cap.reduce_fn = lambda t: None
        cap.gather_fn = lambda out, inp: out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))
        cap.rscatter_fn = lambda out, inp: out.copy_(inp.view(-1)[:out.numel()].view_as(out))
        self._muted = True

        def restore():
            cap.reduce_fn, cap.gather_fn, cap.rscatter_fn = saved
            self._muted = False
        return restore

    def all_reduce_grads(self):
        if self.collectives:
            self.cap.reduce_fn(self.gall)  # the single gradient all-reduce (RCCL over xGMI)

    def train_step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step()

    def capture(self, warmup=2):
        """Capture the whole step into one hipGraph (removes ~300 launch latencies per step).
        Requires fixed shapes and device-generated noise; collectives stay outside graphs."""
        assert not self.collectives and not self.cap.inject
        self.cap.fix_inputs()   # the graph bakes input addresses: later set_batch calls copy INTO these persistent tensors
        if self.fine:
            self.images = self.cap.buf["images"]
        if self.vgg is not None:
            self.vgg.w2_cache = False
            self.vgg.one_stream = True   # (see VggEngine.__init__; the buffers of the one-chain geometry are made by the un-captured steps below)
            warmup = max(warmup, 1)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        self.graph = g

    def eval_rec_loss(self):
        """validate() of main.py:262-284: rec_loss of the TRAINING graph (dropout active, z ~ q)."""
        cap, vgg = self.cap, self.vgg
        feats = vgg.forward(self.images, cap.step) if vgg is not None else None
        cap.forward(feats, train=False)
        return cap.losses()[1]

    def losses(self):
        return self.cap.losses()

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """Every trainable variable of the reference graph (main.py:186-191).  The reference builds the VGG16 variables
        even when it trains on precomputed features (on a dummy input, with the ImageNet weights loaded "for further
        usage", main.py:63-66,205-208, quirk Q22), so its checkpoints always carry cnn/*: without --fine_tune they come
        from `cnn_host` (set by main.py from the ImageNet npz, or kept from the checkpoint that was restored)."""
        d = self.cap.state_dict()
        if self.vgg is not None:
            d.update(self.vgg.state_dict())
        elif self.cnn_host is not None:
            d.update(self.cnn_host)
        return d

    def load_state_dict(self, d, imagenet_path=None):
        self.cap.load_params(d)
        cnn = {k: v for k, v in d.items() if k.startswith("cnn/")}
        if self.vgg is not None:
            if not cnn and imagenet_path and os.path.exists(imagenet_path):
                # a checkpoint written without cnn/* (older runs of this build with precomputed features)
                print("checkpoint has no cnn/* variables: loading VGG16 from %s" % imagenet_path)
                self.vgg.load_weights(imagenet_path)
            else:
                self.vgg.load_params(d)
        elif cnn:
            self.cnn_host = cnn  # carried along so that the next save writes them again

    def save(self, path):
        """saver.save (main.py:286-288): the trainable variables under the reference's names
        (main.py:186-191; optimiser slots and global_step are not in its var list, Q11).
        `path` ending in .npz -> a name-keyed numpy archive; anything else is a TensorFlow V2 checkpoint
        prefix (<path>.index + <path>.data-00000-of-00001 + `checkpoint`, tf_bundle.py)."""
        if path.endswith(".npz"):
            np.savez(path, **self.state_dict())
        else:
            from . import tf_bundle
            tf_bundle.write_bundle(path, self.state_dict())

    def restore(self, path):
        """saver.restore (main.py:201-204, gen_caption.py:113-115): every variable of this model must be present."""
        inet = getattr(self.p, "image_net_weights_path", None)
        if path.endswith(".npz"):
            with np.load(path) as z:
                self.load_state_dict({k: z[k] for k in z.files}, imagenet_path=inet)
        else:
            from . import tf_bundle
            self.load_state_dict(tf_bundle.read_bundle(path), imagenet_path=inet)
