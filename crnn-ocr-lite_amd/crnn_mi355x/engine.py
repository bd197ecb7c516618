"""Device-side engine: owns the flat parameter / gradient / optimizer-state buffers and the workspace
(torch tensors = device memory + streams only) and drives libcrnn_mi355x through its C ABI.

This is what the Keras engine + TensorFlow session were for the reference (Model.fit_generator ->
train_on_batch, train.py:201; Model.predict_generator, predict.py:166).  No CPU fallback exists.
"""
import ctypes
import os
import math
from collections import OrderedDict

import numpy as np
import torch

from . import native
from .native import check, crnn_config


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class StagedBatch:
    """One training batch on its way to the device (Engine.stage): page-locked host buffers -> device buffers on the engine's copy
    stream; `ready` is recorded behind the copies, the compute stream waits for it when the batch is used."""
    __slots__ = ("x", "labels", "input_length", "label_length", "ready", "slot", "n")

    def __len__(self):
        return self.n


class Engine:
    def __init__(self, batch, imgh=100, imgw=32, num_classes=38, max_len=23, time_dense_size=128, n_units=256,
                 gru=False, stn=True, dropout=True, device=None, precision="fp32", share=None, flags=None):
        """share: another Engine of the same architecture (any batch size) whose parameter / gradient / BatchNorm / optimizer-
        state tensors this one adopts (only the workspace and the batch-shaped outputs are its own).
        precision: "fp32" = the PARITY mode -- fp32 tensors; FORWARD GEMMs with fp32-accurate products (three bf16 planes per operand: what the 1e-3 logit /
        CTC-loss tolerance and the bit-exact arg-max are asserted on); BACKWARD GEMMs (pointwise convolutions, dense layers, RNN projections -- not the
        recurrences) with TWO planes per operand = 16 significant bits per factor (a product's relative error <= 3 * 2^-18; against the three-plane backward the
        gradients agree to 1e-4 of the whole gradient's L2 norm and, tensor by tensor, to 1e-3 of the tensor's largest element -- the bounds the tests assert;
        measured 8e-6 and 1.3e-4 -- and both are checked against the fp64 oracle at batch 64).  flags=native.FLAG_THREE_PLANE_BACKWARD is the strict form (three
        planes in the backward too, about 10 % slower); FLAG_TWO_PLANE_FORWARD (opt-in) narrows the forward as well.  "bf16" = bf16 MFMA products, fp32
        tensors; "bf16s" = additionally bf16 conv-stack tensors (the throughput mode; outside the 1e-3 tolerance).
        flags: bit set of native.FLAG_* (default: $CRNN_FLAGS or 0): schedule A/B switches -- same results bit for bit or to summation order, as
        include/crnn_mi355x.h says per flag -- and the parity mode's product-precision switches FLAG_THREE_PLANE_BACKWARD / FLAG_TWO_PLANE_FORWARD /
        FLAG_F32_MFMA_GEMMS; round 6: FLAG_NO_GRADIENT_PLANES (BatchNorm-2's input gradients as fp32 tensors instead of bf16 planes) and
        FLAG_NO_POOL_ARGMAX_Q (pooled blocks: window scan instead of the saved arg-max values)."""
        if not torch.cuda.is_available():
            raise RuntimeError("the CRNN hot path needs an AMD GPU (gfx950); there is no CPU fallback")
        self.lib = native.lib()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.cfg = crnn_config(batch, imgh, imgw, num_classes, max_len, time_dense_size, n_units, int(bool(gru)),
                               int(bool(stn)), int(bool(dropout)), {"fp32": 0, "bf16": 1, "bf16s": 2}.get(precision, 0),
                               int(os.environ.get("CRNN_FLAGS", "0")) if flags is None else int(flags))
        if precision not in ("fp32", "bf16", "bf16s"):
            raise ValueError("precision must be 'fp32' (parity mode), 'bf16' (bf16 MFMA products, fp32 tensors) or "
                             "'bf16s' (bf16 MFMA products + bf16 conv-stack tensors in HBM)")
        self.precision = precision
        self._c = ctypes.byref(self.cfg)
        nbytes = self.lib.crnn_workspace_bytes(self._c)
        if nbytes == 0:
            raise native.CrnnError("unsupported CRNN configuration (n_units % 64 == 0, classes <= 64, max_len <= 31 required)")
        self.T = self.lib.crnn_time_steps(self._c)
        self.B, self.C = batch, num_classes
        self.n_total = self.lib.crnn_params_total(self._c)
        self.bn_total = self.lib.crnn_bn_total(self._c)
        self.layout = OrderedDict()
        name = ctypes.create_string_buffer(64)
        off, size, ndim = ctypes.c_long(), ctypes.c_long(), ctypes.c_int()
        dims = (ctypes.c_int * 4)()
        for i in range(self.lib.crnn_num_params(self._c)):
            check(self.lib.crnn_param_info(self._c, i, name, 64, ctypes.byref(off), ctypes.byref(size), ctypes.byref(ndim), dims))
            self.layout[name.value.decode()] = (off.value, size.value, tuple(dims[:ndim.value]))
        self.bn_layout = OrderedDict()
        boff, bch, bcnt = ctypes.c_int(), ctypes.c_int(), ctypes.c_long()
        for i in range(14):
            check(self.lib.crnn_bn_info(self._c, i, name, 64, ctypes.byref(boff), ctypes.byref(bch), ctypes.byref(bcnt)))
            self.bn_layout[name.value.decode()] = (boff.value, bch.value, bcnt.value)
        dev = self.device
        if share is not None:
            if share.n_total != self.n_total or share.bn_total != self.bn_total or list(share.layout.items()) != list(self.layout.items()):
                raise ValueError("Engine(share=...) needs the same architecture")
            if share.device != self.device or share.precision != precision:
                raise ValueError("Engine(share=...) needs the same device and precision (%s/%s vs %s/%s): the parameter, gradient and "
                                 "optimizer-state tensors are shared" % (share.device, share.precision, self.device, precision))
            self.params, self.grads, self.bn_mean, self.bn_var = share.params, share.grads, share.bn_mean, share.bn_var
        else:
            self.params = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
            self.grads = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
            self.bn_mean = torch.zeros(self.bn_total, dtype=torch.float32, device=dev)
            self.bn_var = torch.ones(self.bn_total, dtype=torch.float32, device=dev)
        self.ws_bytes = nbytes
        # zeroed once: the sticky give-up counter of the persistent recurrences lives in it (include/crnn_mi355x.h, "rnnx") and no
        # launch ever resets that word
        self.ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        self._rnn_giveups = None          # 1-element int32 view of that counter (None: this configuration runs the per-step kernels)
        self._rnn_giveups_seen = 0        # baseline of this rank's counter
        self._rnn_giveups_seen_sum = 0    # baseline of the counter summed over data-parallel ranks (surface.fit_generator)
        off, cnt = ctypes.c_long(), ctypes.c_long()
        if self.lib.crnn_ws_tensor(self._c, b"rnnx", ctypes.byref(off), ctypes.byref(cnt)) == 0:
            self._rnn_giveups = self.ws[off.value:off.value + 1].view(torch.int32)
        self.y_pred = torch.empty((batch, self.T, num_classes), dtype=torch.float32, device=dev)
        self.loss = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self.norm_scratch = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.opt_state = share.opt_state if share is not None else {}     # Adam m/v or SGD velocity: survives a batch-size change
        # side-stream schedule of the backward (crnn_backward_ex): measured 1 % SLOWER than the serial one at batch 256 (the
        # GEMMs crowd the latency-critical BPTT launches), so it is opt-in: CRNN_RNN_OVERLAP=1
        self.overlap_rnn_wgrad = os.environ.get("CRNN_RNN_OVERLAP", "0") == "1"
        # conv stack (crnn_backward_bottom_ex): the pointwise weight-gradient GEMM of each block on the side stream next to the block's
        # bandwidth-bound kernels -- measured neutral to 0.7 % slower at batch 256 (7.97 vs 7.91 ms: both sides already fill the CUs),
        # so also opt-in: CRNN_CONV_OVERLAP=1; bit-identical either way
        self.overlap_conv_wgrad = os.environ.get("CRNN_CONV_OVERLAP", "0") == "1"
        self.overlap_keep_bytes = os.environ.get("CRNN_KEEP_OVERLAP", "1") == "1"
        self._aux_stream = None
        self._warn_schedule_fallbacks()
        self._stage = None                # page-locked staging slots of stage() (built on first use)
        self._stage_next = 0

    _warned_shapes = set()

    def _warn_schedule_fallbacks(self):
        """The row-stream depthwise kernels (csrc/dwconv_stream.hip, dwconv_bwd_stream.hip) take maps whose rows (W * C * 2 bytes, or a
        whole number of row bands) fill their 9 KiB step row -- every block at image width 32.  Other widths run the halo-tile kernels
        at roughly half the depthwise rate: correct, but worth knowing about, so say it once per shape instead of silently."""
        if self.precision != "bf16s" or (self.cfg.flags & native.FLAG_DW_TILE_KERNEL):
            return
        key = (self.cfg.imgh, self.cfg.imgw)
        if key in Engine._warned_shapes:
            return
        h, w, cin = self.cfg.imgh + 4, self.cfg.imgw + 4, 1
        slow = []
        for i, (co, ph, pw) in enumerate(((64, 1, 1), (128, 1, 1), (256, 2, 2), (256, 1, 1), (512, 1, 2), (512, 1, 1), (512, 1, 1)), 1):
            if i >= 2 and (self.lib.crnn_dwconv_fwd_stream_supported(self.B, h, w, cin) != 0 or self.lib.crnn_dwconv_bwd_stream_supported(self.B, h, w, cin) != 0):
                slow.append("block %d (%dx%dx%d)" % (i, h, w, cin))
            h, w, cin = h // ph, w // pw, co
        if slow:
            import warnings
            Engine._warned_shapes.add(key)
            warnings.warn("crnn_mi355x: image width %d: the row-stream depthwise kernels do not take %s (their step row is sized for width 32); "
                          "these blocks run the halo-tile kernels at about half the depthwise rate" % (self.cfg.imgw, ", ".join(slow)), RuntimeWarning, stacklevel=3)

    # ---- host -> device staging --------------------------------------------------------------------------
    _STAGE_SLOTS = 2

    def _stage_slots(self):
        if self._stage is None:
            B, c = self.B, self.cfg
            slots = []
            for _ in range(self._STAGE_SLOTS):
                s = {"hx": torch.empty(B * c.imgh * c.imgw, dtype=torch.float32).pin_memory(),
                     "hi": torch.empty(B * (c.max_len + 2), dtype=torch.int32).pin_memory(),
                     "dx": torch.empty(B * c.imgh * c.imgw, dtype=torch.float32, device=self.device),
                     "di": torch.empty(B * (c.max_len + 2), dtype=torch.int32, device=self.device),
                     "copied": torch.cuda.Event(), "free": torch.cuda.Event()}
                s["hx_np"], s["hi_np"] = s["hx"].numpy(), s["hi"].numpy()
                slots.append(s)
            self._stage = slots
            self._copy_stream = torch.cuda.Stream(device=self.device)
        return self._stage

    def stage(self, x, labels, input_length, label_length):
        """Host batch (Readf's NumPy arrays: float64 images, int64 labels / lengths) -> StagedBatch.  The images are converted to
        fp32 straight into a page-locked buffer (one pass: the snapshot the generator contract asks for -- Readf keeps writing into the
        arrays it yielded, utils.py:468,495-511 -- and the cast in one), labels and lengths into a second one, and both are copied to
        the device asynchronously on a copy stream: staging batch k+1 while step k runs hides the PCIe transfer (train.py:201-209:
        fit_generator over a host generator).  Two slots: a slot is rewritten only after the step that used it has finished."""
        slots = self._stage_slots()
        k = self._stage_next
        s = slots[k]
        B, c = self.B, self.cfg
        xa = np.asarray(x)
        # validation first: a rejected batch leaves the slot ring and the staging buffers untouched
        if xa.size != s["hx_np"].size:
            raise ValueError("batch shape mismatch: the_input has %d elements, the engine expects %d x %d x %d" % (xa.size, B, c.imgh, c.imgw))
        self._check_ctc_inputs(labels, input_length, label_length)
        self._stage_next = (k + 1) % len(slots)
        s["copied"].synchronize()                      # the previous H->D copy out of this slot's host buffers is done
        with np.errstate(over="ignore", invalid="ignore"):
            np.copyto(s["hx_np"], xa.reshape(-1), casting="unsafe")
        if not np.isfinite(s["hx_np"]).all():         # undefined rows of Readf's short tail batch (see _as_input)
            s["hx_np"][~np.isfinite(s["hx_np"])] = 0.0
        L = c.max_len
        hi = s["hi_np"]
        np.copyto(hi[:B * L], np.asarray(labels).reshape(-1), casting="unsafe")
        np.copyto(hi[B * L:B * L + B], np.asarray(input_length).reshape(-1), casting="unsafe")
        np.copyto(hi[B * L + B:], np.asarray(label_length).reshape(-1), casting="unsafe")
        cs = self._copy_stream
        cs.wait_event(s["free"])                      # the step that last read this slot's device buffers has finished
        with torch.cuda.stream(cs):
            s["dx"].copy_(s["hx"], non_blocking=True)
            s["di"].copy_(s["hi"], non_blocking=True)
            s["copied"].record(cs)
        sb = StagedBatch()
        sb.x, sb.ready, sb.slot, sb.n = s["dx"], s["copied"], s, B
        sb.labels, sb.input_length, sb.label_length = s["di"][:B * L], s["di"][B * L:B * L + B], s["di"][B * L + B:]
        return sb

    def _release(self, sb):
        """The compute stream has been handed every kernel that reads the staged batch: its slot may be rewritten after them."""
        sb.slot["free"].record(torch.cuda.current_stream())

    # ---- parameters -------------------------------------------------------------------------------------
    def set_params(self, p, bn=None):
        """p: {name: ndarray} in the oracle/Keras naming (oracle.model.Config.param_shapes)."""
        flat = np.zeros(self.n_total, dtype=np.float32)
        for name, (off, size, dims) in self.layout.items():
            a = np.asarray(p[name], dtype=np.float32)
            assert a.size == size, (name, a.shape, dims)
            flat[off:off + size] = a.reshape(-1)
        self.params.copy_(torch.from_numpy(flat))
        if bn is not None:
            m = np.zeros(self.bn_total, np.float32); v = np.ones(self.bn_total, np.float32)
            for name, (off, ch, _) in self.bn_layout.items():
                m[off:off + ch] = bn[name + "_mean"]; v[off:off + ch] = bn[name + "_var"]
            self.bn_mean.copy_(torch.from_numpy(m)); self.bn_var.copy_(torch.from_numpy(v))

    def _unflatten(self, flat):
        flat = flat.detach().cpu().numpy()
        return OrderedDict((n, flat[o:o + s].reshape(d).copy()) for n, (o, s, d) in self.layout.items())

    def get_params(self):
        return self._unflatten(self.params)

    def get_grads(self):
        return self._unflatten(self.grads)

    def get_bn(self):
        m, v = self.bn_mean.cpu().numpy(), self.bn_var.cpu().numpy()
        out = OrderedDict()
        for name, (off, ch, _) in self.bn_layout.items():
            out[name + "_mean"] = m[off:off + ch].copy(); out[name + "_var"] = v[off:off + ch].copy()
        return out

    def ws_tensor(self, name):
        """Named view into the workspace, in its storage type (torch.float32 or torch.bfloat16)."""
        off, cnt, dt = ctypes.c_long(), ctypes.c_long(), ctypes.c_int()
        check(self.lib.crnn_ws_tensor_info(self._c, name.encode(), ctypes.byref(off), ctypes.byref(cnt), ctypes.byref(dt)), "ws_tensor " + name)
        if dt.value == 1:
            return self.ws[off.value:off.value + (cnt.value + 1) // 2].view(torch.bfloat16)[:cnt.value]
        return self.ws[off.value:off.value + cnt.value]

    # ---- hot path -----------------------------------------------------------------------------------------
    def _as_input(self, x):
        if isinstance(x, StagedBatch):
            torch.cuda.current_stream().wait_event(x.ready)
            return x.x
        if not torch.is_tensor(x):
            # Readf's batches start as np.empty (utils.py:448) and the short tail batch of a pass is yielded with
            # whatever those unused rows contain: rows that are not finite in fp32 are zeroed instead of poisoning
            # the batch statistics of every other image (the values are undefined in the reference as well)
            with np.errstate(over="ignore", invalid="ignore"):
                x = np.ascontiguousarray(x, dtype=np.float32)
            bad = ~np.isfinite(x)
            if bad.any():
                x = np.where(bad, np.float32(0), x)
            x = torch.from_numpy(x)
        x = x.to(self.device, dtype=torch.float32, non_blocking=True).contiguous()
        assert x.numel() == self.B * self.cfg.imgh * self.cfg.imgw, "batch shape mismatch"
        return x

    def forward(self, x, train=False, seed=0):
        """x (B,imgh,imgw,1) -> y_pred (B,T,C) device tensor (softmax)."""
        x = self._as_input(x)
        self._x = x
        # (training: the dropout keep bytes of the fused block outputs are generated on the side stream next to the spatial transformer)
        check(self.lib.crnn_forward_ex(self._c, _ptr(self.params), _ptr(self.bn_mean), _ptr(self.bn_var), _ptr(x), _ptr(self.ws),
                                       self.ws_bytes, _ptr(self.y_pred), int(train), int(seed), _stream(), self._aux(bool(train) and self.overlap_keep_bytes)), "forward")
        return self.y_pred

    def _as_i32(self, a):
        if not torch.is_tensor(a):
            a = torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int32))
        return a.to(self.device, dtype=torch.int32).contiguous()

    def _ctc_inputs(self, labels, input_length, label_length):
        if isinstance(labels, StagedBatch):           # validated on the host by stage(); already on the device
            torch.cuda.current_stream().wait_event(labels.ready)
            self._lab, self._il, self._ll = labels.labels, labels.input_length, labels.label_length
            return
        self._check_ctc_inputs(labels, input_length, label_length)
        self._lab = self._as_i32(labels); self._il = self._as_i32(input_length); self._ll = self._as_i32(label_length)

    def _check_ctc_inputs(self, labels, input_length, label_length):
        """Host-side validation of the CTC inputs (the kernel indexes LDS with the label ids): labels (B, max_len) with
        0 <= id < num_classes, 0 <= label_length <= max_len, input_length <= T - 2.  Device tensors are trusted (checking
        them would force a synchronisation in the hot loop); NumPy batches from Readf are checked."""
        if not torch.is_tensor(labels):
            lab = np.asarray(labels)
            if lab.size != self.B * self.cfg.max_len:
                raise ValueError("the_labels must be (batch=%d, max_len=%d), got %r" % (self.B, self.cfg.max_len, lab.shape))
            if lab.size and (lab.min() < 0 or lab.max() >= self.C):
                raise ValueError("label ids must be in [0, %d) (blank = %d), got [%d, %d]" % (self.C, self.C - 1, lab.min(), lab.max()))
        if not torch.is_tensor(label_length):
            ll = np.asarray(label_length)
            if ll.size != self.B or (ll.size and (ll.min() < 0 or ll.max() > self.cfg.max_len)):
                raise ValueError("label_length must be (batch,) with values in [0, max_len=%d]" % self.cfg.max_len)
        if not torch.is_tensor(input_length):
            il = np.asarray(input_length)
            if il.size != self.B or (il.size and (il.min() < 0 or il.max() > self.T - 2)):
                raise ValueError("input_length must be (batch,) with values in [0, T-2=%d]" % (self.T - 2))

    def backward(self, labels, input_length, label_length, seed=0):
        """CTC + backward after forward(train=True).  Returns per-sample loss (device tensor, B)."""
        self.backward_top(labels, input_length, label_length, seed=seed)
        self.backward_bottom(seed=seed)
        return self.loss

    def backward_top(self, labels, input_length, label_length, seed=0):
        """First backward stage: CTC, dense2, recurrent layers, dense1 -> grads[grad_split:] are final."""
        self._ctc_inputs(labels, input_length, label_length)
        check(self.lib.crnn_backward_top_ex(self._c, _ptr(self.params), _ptr(self.grads), _ptr(self._lab), _ptr(self._il), _ptr(self._ll),
                                            _ptr(self.ws), self.ws_bytes, _ptr(self.loss), int(seed), _stream(), self._aux(self.overlap_rnn_wgrad)), "backward_top")
        return self.loss

    def backward_bottom(self, seed=0):
        """Second backward stage: conv stack + spatial transformer -> grads[:grad_split]."""
        check(self.lib.crnn_backward_bottom_ex(self._c, _ptr(self.params), _ptr(self.grads), _ptr(self._x), _ptr(self.ws), self.ws_bytes,
                                               int(seed), _stream(), self._aux(self.overlap_conv_wgrad)), "backward_bottom")

    @property
    def grad_split(self):
        return int(self.lib.crnn_grad_split_offset(self._c))

    def _aux(self, on):
        """Second stream for the weight-gradient GEMMs that run next to the backward's critical path (None: serial schedule)."""
        if not on:
            return None
        if self._aux_stream is None:
            self._aux_stream = torch.cuda.Stream(device=self.device)
        return ctypes.c_void_p(self._aux_stream.cuda_stream)

    def bn_update(self):
        check(self.lib.crnn_bn_update(self._c, _ptr(self.bn_mean), _ptr(self.bn_var), _ptr(self.ws), self.ws_bytes, _stream()), "bn_update")

    def global_norm(self, clipnorm):
        check(self.lib.crnn_global_norm(_ptr(self.grads), self.n_total, float(clipnorm or 0.0), _ptr(self.norm_scratch), _ptr(self.norm),
                                        _stream()), "global_norm")
        return self.norm

    def adam_step(self, lr, beta_1, beta_2, epsilon, clipnorm, iteration):
        """Keras 2.2.2 Adam; `iteration` is 0-based."""
        st = self.opt_state
        if "m" not in st:
            st["m"] = torch.zeros_like(self.params); st["v"] = torch.zeros_like(self.params)
        self.global_norm(clipnorm)
        t = iteration + 1
        lr_t = lr * math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
        check(self.lib.crnn_adam_step(_ptr(self.params), _ptr(self.grads), _ptr(st["m"]), _ptr(st["v"]), self.n_total, lr_t, beta_1,
                                      beta_2, epsilon, _ptr(self.norm), _stream()), "adam")

    def sgd_step(self, lr, decay, momentum, nesterov, clipnorm, iteration):
        st = self.opt_state
        if "vel" not in st:
            st["vel"] = torch.zeros_like(self.params)
        self.global_norm(clipnorm)
        lr_i = lr / (1.0 + decay * iteration)
        check(self.lib.crnn_sgd_step(_ptr(self.params), _ptr(self.grads), _ptr(st["vel"]), self.n_total, lr_i, momentum, int(nesterov),
                                     _ptr(self.norm), _stream()), "sgd")

    def train_step(self, x, labels, input_length, label_length, opt, iteration, seed=None, allreduce=None):
        """forward(train) -> CTC -> backward -> [all-reduce] -> clip -> optimizer -> BN moving stats.
        x may be a StagedBatch (stage()): it then carries the labels and lengths too."""
        seed = iteration if seed is None else seed
        staged = x if isinstance(x, StagedBatch) else None
        if staged is not None:
            labels = staged
        try:
            self.forward(x, train=True, seed=seed)
            if allreduce is not None and getattr(allreduce, "overlap", False):
                # the upper layers' gradients (tail of the flat buffer) are exchanged while the conv stack is still in backward
                loss = self.backward_top(labels, input_length, label_length, seed=seed)
                split = self.grad_split
                allreduce.start(self.grads[split:])
                self.backward_bottom(seed=seed)
                allreduce.start(self.grads[:split])
                allreduce.finish(self.grads)
            else:
                loss = self.backward(labels, input_length, label_length, seed=seed)
                if allreduce is not None:
                    allreduce(self.grads)
        finally:
            if staged is not None:
                # also when a launch raised: `free` must follow whatever was enqueued on the slot's buffers, or the next H->D copy
                # into the slot would wait on a stale event and could race kernels still reading it
                self._release(staged)
        opt.apply(self, iteration)
        self.bn_update()
        return loss

    # ---- persistent-recurrence status ------------------------------------------------------------------------
    def loss_and_status(self):
        """Device tensor [mean CTC cost of the last step, give-up counter]: ONE small D2H copy serves the loss read-back of the
        training loop and the status check (`raise_if_rnn_gave_up(values[1])`), no extra synchronisation."""
        st = self._rnn_giveups.to(torch.float32) if self._rnn_giveups is not None else torch.zeros(1, device=self.device)
        return torch.cat([self.loss.mean().reshape(1), st])

    def raise_if_rnn_gave_up(self, count, summed=False):
        """count: current value of the sticky give-up counter (from loss_and_status / check_rnn_status).  A persistent recurrence
        whose cluster was not co-resident (CUs held by another process / stream, a CU-masked device) stops waiting after a bounded
        spin and free-runs on garbage; that must never pass silently.  summed: `count` is the SUM of the ranks' counters (data
        parallel); it has its own baseline, so the rank-local checks of predict / test never see a value of the other kind."""
        count = int(count)
        seen = self._rnn_giveups_seen_sum if summed else self._rnn_giveups_seen
        if count != seen:
            new = count - seen
            if summed:
                self._rnn_giveups_seen_sum = count
            else:
                self._rnn_giveups_seen = count
            raise native.CrnnError("persistent LSTM/GRU recurrence gave up waiting for its workgroup cluster %d time(s): the launch was not "
                                   "co-resident on the GPU, so the results of this step (and weights updated from it) are invalid.  Free the GPU "
                                   "of other work, or run the per-step recurrence kernels: Engine(flags=native.FLAG_RNN_STEP_KERNELS) / "
                                   "CRNN_FLAGS=1" % new)

    def check_rnn_status(self):
        """Synchronising check (4-byte D2H) for call sites that already wait for the device (predict, validation)."""
        if self._rnn_giveups is not None:
            self.raise_if_rnn_gave_up(int(self._rnn_giveups.item()))

    # ---- decoding -----------------------------------------------------------------------------------------
    def greedy_decode(self, y=None, input_length=None):
        y = self.y_pred if y is None else y
        B, T, C = y.shape
        out = torch.empty((B, T), dtype=torch.int32, device=self.device); ln = torch.empty(B, dtype=torch.int32, device=self.device)
        il = self._as_i32(input_length) if input_length is not None else None
        check(self.lib.crnn_ctc_greedy_decode(_ptr(y), _ptr(il), _ptr(out), _ptr(ln), B, T, C, _stream()), "greedy")
        return out, ln

    def beam_decode(self, y=None, beam_width=10, merge_repeated=True, input_length=None):
        y = self.y_pred if y is None else y
        return beam_decode(y, beam_width, merge_repeated, input_length)


def beam_decode(y, beam_width=10, merge_repeated=True, input_length=None):
    """y (B,T,C) softmax (device tensor or ndarray) -> (labels (B,T) int32 padded -1, lengths, scores)."""
    lib = native.lib()
    if not torch.is_tensor(y):
        y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).cuda()
    y = y.contiguous().float()
    B, T, C = y.shape
    out = torch.empty((B, T), dtype=torch.int32, device=y.device); ln = torch.empty(B, dtype=torch.int32, device=y.device)
    sc = torch.empty(B, dtype=torch.float32, device=y.device)
    il = None
    if input_length is not None:
        il = torch.as_tensor(np.asarray(input_length).reshape(-1).astype(np.int32)).to(y.device)
    check(lib.crnn_ctc_beam_decode(_ptr(y), _ptr(il), _ptr(out), _ptr(ln), _ptr(sc), B, T, C, int(beam_width), int(bool(merge_repeated)),
                                   _stream()), "beam_decode")
    return out, ln, sc
