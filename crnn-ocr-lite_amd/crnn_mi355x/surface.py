"""The reference's model-facing Python surface, re-hosted on the HIP engine.

Same names / signatures / return types as utils.py (CRNN, init_predictor, load_custom_model, save_model_json,
load_model_custom, ctc_lambda_func, BilinearInterpolation, STN) and the subset of the Keras `Model` API that
train.py / predict.py / EarlyStoppingIter actually touch (compile, fit_generator, predict_generator,
evaluate_generator, get/set_weights, load/save_weights, save, to_json, summary, stop_training, get_layer,
input).  The model is not a graph of Python layer objects: `get_model()` returns a handle whose train /
predict steps are single calls into libcrnn_mi355x (forward + CTC + backward + optimizer as HIP kernels).

Storage: `save_weights` / `save` / `load_weights` read and write real HDF5 files in the layout Keras 2.2.2 uses
(`layer_names` / `weight_names` attributes, one group per layer, one contiguous float32 dataset per weight; `save` puts
them under `/model_weights` next to a `model_config` attribute) through the package's own HDF5 subset (hdf5.py; h5py is
not importable here).  Paths ending in `.npz` -- and archives written by earlier revisions under `.h5` names -- are
NumPy archives in Keras' weight order.
"""
import ctypes
import json
import os
import time
from collections import OrderedDict

import numpy as np

from . import hdf5, native
from .init import initial_parameters

BLOCK_FILTERS = (64, 128, 256, 256, 512, 512, 512)
BLOCK_POOL = (None, None, (2, 2), None, (1, 2), None, None)


def _cfg_struct(batch, shape, num_classes, max_len, tds, units, gru, stn=True, dropout=True, mfma_bf16=False):
    return native.crnn_config(int(batch), int(shape[0]), int(shape[1]), int(num_classes), int(max_len), int(tds), int(units),
                              int(bool(gru)), int(bool(stn)), int(bool(dropout)), int(bool(mfma_bf16)))


def param_layout(cfg):
    """{name: (offset, size, dims)} from the C ABI (host-only call, no GPU needed)."""
    lib = native.lib()
    c = ctypes.byref(cfg)
    name = ctypes.create_string_buffer(64)
    off, size, ndim = ctypes.c_long(), ctypes.c_long(), ctypes.c_int()
    dims = (ctypes.c_int * 4)()
    out = OrderedDict()
    for i in range(lib.crnn_num_params(c)):
        native.check(lib.crnn_param_info(c, i, name, 64, ctypes.byref(off), ctypes.byref(size), ctypes.byref(ndim), dims))
        out[name.value.decode()] = (off.value, size.value, tuple(dims[:ndim.value]))
    return out


class _Tensor:
    """Symbolic stand-in for `model.input` / `layer.output` (only identity matters to the callers)."""

    def __init__(self, name, shape):
        self.name, self.shape = name, shape

    def __repr__(self):
        return "<tensor %s %s>" % (self.name, self.shape)


class _Layer:
    def __init__(self, name, output):
        self.name, self.output = name, output


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []


class CRNN:
    """utils.py:32-96 -- same constructor; `get_model()` returns the trainable 4-input model whose single output
    is the per-sample CTC cost ('ctc')."""

    def __init__(self, num_classes=97, max_string_len=23, shape=(40, 40, 1), time_dense_size=128, GRU=False, n_units=256):
        self.num_classes = num_classes
        self.shape = shape
        self.max_string_len = max_string_len
        self.n_units = n_units
        self.GRU = GRU
        self.time_dense_size = time_dense_size

    def get_model(self):
        self.pooling_counter_h, self.pooling_counter_w = 0, 0
        for pool in BLOCK_POOL:                      # utils.py:50-55
            if pool is not None:
                self.pooling_counter_h += int(pool[0] == 2)
                self.pooling_counter_w += int(pool[1] == 2)
        return Model(dict(num_classes=self.num_classes, max_string_len=self.max_string_len, shape=tuple(self.shape),
                          time_dense_size=self.time_dense_size, GRU=bool(self.GRU), n_units=self.n_units))


class Model:
    def __init__(self, config, predictor=False, share=None):
        self.config = dict(config)
        self.predictor = predictor
        self.stop_training = False
        self.optimizer = None
        self._iterations = 0
        self._prefetched = None          # (generator, batch) drawn ahead of time by fit_generator (kept across epochs)
        self._keras_json = None          # the Keras model.json this model was built from (layer names for HDF5 files)
        sh = self.config["shape"]
        self._T = (sh[0] + 4) // 2
        if share is not None:                        # predictor view of an existing model: same weights/engine
            self._state = share._state
        else:
            cfg = _cfg_struct(1, sh, self.config["num_classes"], self.config["max_string_len"], self.config["time_dense_size"],
                              self.config["n_units"], self.config["GRU"])
            layout = param_layout(cfg)
            p = initial_parameters(layout, self.config["n_units"], self.config["GRU"])
            bn = OrderedDict()
            cin = 1
            for i, co in enumerate(BLOCK_FILTERS, 1):
                for j, ch in ((1, cin), (2, co)):
                    bn["b%d_bn%d_mean" % (i, j)] = np.zeros(ch, np.float32)
                    bn["b%d_bn%d_var" % (i, j)] = np.ones(ch, np.float32)
                cin = co
            self._state = {"params": p, "bn": bn, "engine": None, "layout": layout}
        C = self.config["num_classes"]
        self.input = [_Tensor("the_input", (None,) + tuple(sh)), _Tensor("the_labels", (None, self.config["max_string_len"])),
                      _Tensor("input_length", (None, 1)), _Tensor("label_length", (None, 1))]
        self._layers = OrderedDict((n, _Layer(n, _Tensor(n, s))) for n, s in (
            ("the_input", (None,) + tuple(sh)), ("softmax", (None, self._T, C)), ("ctc", (None, 1))))

    # ---- engine management --------------------------------------------------------------------------------
    def _engine(self, batch, dropout=True):
        """Engine for this batch size.  Engines differ only in their workspace: the parameter, gradient, BatchNorm and
        optimizer-state tensors are shared between them, so a short tail batch or a validation pass with another batch size
        neither resets Adam's moments nor round-trips the weights through the host.  The three most recent sizes are kept."""
        from .engine import Engine
        st = self._state
        engines = st.setdefault("engines", OrderedDict())
        eng = engines.get(batch)
        if eng is None:
            base = st["engine"]
            c = self.config
            eng = Engine(batch, c["shape"][0], c["shape"][1], c["num_classes"], c["max_string_len"], c["time_dense_size"], c["n_units"],
                         gru=c["GRU"], stn=True, dropout=dropout, precision=os.environ.get("CRNN_PRECISION", "fp32"), share=base)
            if base is None:
                eng.set_params(st["params"], st["bn"])     # no collective here: see sync_replicas()
            engines[batch] = eng
            while len(engines) > 3:
                engines.popitem(last=False)
        else:
            engines.move_to_end(batch)
        st["engine"] = eng
        return eng

    def sync_replicas(self, batch=None):
        """Data parallel, a COLLECTIVE (every rank must call it): all ranks adopt rank 0's weights / BatchNorm moving statistics
        (a fresh model is initialised from OS entropy per process; crnn_mi355x.parallel.broadcast_state).  fit_generator calls it once
        when it starts.  set_weights / load_weights and the lazy building of an engine are rank-local and issue no collective, so
        rank-conditional user code (`if rank == 0: model.predict_on_batch(...)`, `if rank == 0: model.load_weights(...)`) cannot
        deadlock the other ranks by itself; they only mark THIS rank's replica as out of sync.  Whether a broadcast is needed is
        then decided COLLECTIVELY, never from the rank-local mark alone: a data-parallel `train_on_batch` first all-reduces (MAX)
        the ranks' marks (`_replicas_need_sync`, one 4-byte collective every rank takes part in) and all ranks broadcast, or none
        does -- after `if rank == 0: model.load_weights(...)` every rank therefore adopts the loaded weights at the next train
        step instead of rank 0 entering a broadcast the others never join.  fit_generator's pipelined loop skips that per-step
        check (it would cost a host synchronisation per step): it synchronises once at the start, and carries every rank's mark in
        a third slot of the loss all-reduce it issues anyway -- weights replaced on one rank by a callback are re-broadcast from
        rank 0 after the following step (round 6)."""
        dist, world = self._dist()
        if world > 1:
            from .parallel import broadcast_state
            eng = self._state.get("engine")
            if eng is None:
                eng = self._engine(batch or 1)
            broadcast_state(eng, dist, world)
            self._state["replicas_synced"] = True

    def _replicas_need_sync(self, eng, dist):
        """COLLECTIVE: True on every rank iff any rank's replica is marked out of sync (fresh model, set_weights, load_weights).
        The mark is rank-local; the decision to broadcast must not be (one rank in a broadcast, the others in the gradient
        all-reduce = mismatched collectives)."""
        import torch
        flag = torch.tensor([0 if self._state.get("replicas_synced") else 1], dtype=torch.int32, device=eng.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return bool(int(flag.item()))

    def _pull(self):
        """device -> host copies of weights and BN statistics."""
        eng = self._state["engine"]
        if eng is not None:
            self._state["params"] = eng.get_params()
            self._state["bn"] = eng.get_bn()

    def _push(self):
        eng = self._state["engine"]
        if eng is not None:
            eng.set_params(self._state["params"], self._state["bn"])     # rank-local (see sync_replicas)
        self._state["replicas_synced"] = False       # the next data-parallel train step (or fit_generator) re-broadcasts rank 0's

    # ---- Keras weight list (SURVEY A.9 order, 94 tensors for either cell) ---------------------------------------
    def _weight_index(self):
        idx = [("p", n) for n in ("stn_c1_k", "stn_c1_b", "stn_c2_k", "stn_c2_b", "stn_d1_w", "stn_d1_b", "stn_d2_w", "stn_d2_b")]
        for i in range(1, 8):
            b = "b%d" % i
            idx += [("p", b + "_dw"), ("p", b + "_bn1_g"), ("p", b + "_bn1_b"), ("s", b + "_bn1_mean"), ("s", b + "_bn1_var"),
                    ("p", b + "_pw"), ("p", b + "_bn2_g"), ("p", b + "_bn2_b"), ("s", b + "_bn2_mean"), ("s", b + "_bn2_var")]
        idx += [("p", "dense1_w"), ("p", "dense1_b")]
        for l in (1, 2):
            for d in ("f", "b"):
                idx += [("p", "rnn%d%s_%s" % (l, d, k)) for k in ("w", "u", "b")]
        idx += [("p", "dense2_w"), ("p", "dense2_b")]
        return idx

    @staticmethod
    def _keras_shape(name, a):
        if name.endswith("_dw"):
            return a.reshape(a.shape + (1,))                 # (3,3,C,1)
        if name.endswith("_pw"):
            return a.reshape((1, 1) + a.shape)               # (1,1,Cin,Cout)
        return a

    def get_weights(self):
        self._pull()
        st = self._state
        return [self._keras_shape(n, np.array(st["params"][n] if k == "p" else st["bn"][n])) for k, n in self._weight_index()]

    def set_weights(self, weights):
        st = self._state
        idx = self._weight_index()
        if len(weights) != len(idx):
            raise ValueError("expected %d weight arrays, got %d" % (len(idx), len(weights)))
        for (k, n), w in zip(idx, weights):
            tgt = st["params"] if k == "p" else st["bn"]
            tgt[n] = np.asarray(w, dtype=np.float32).reshape(tgt[n].shape)
        self._push()

    # ---- Keras HDF5 layout (keras/engine/saving.py of 2.2.2: save_weights_to_hdf5_group / load_weights_from_hdf5_group) ----
    def _keras_layers(self):
        """[(layer name, [(weight name, ('p'|'s', engine tensor name))])] for every layer of the Keras graph, in
        `model.layers` order.  Names come from the Keras model.json this model was built from, else they are the names a
        fresh Keras session gives CRNN.get_model() (= those in the reference's models/*/model.json)."""
        widx = iter(self._weight_index())
        take = lambda names: [(n, next(widx)) for n in names]
        if self._keras_json is not None:
            specs, nbi = [], 0
            for l in self._keras_json["config"]["layers"]:
                if l["class_name"] == "Bidirectional":
                    nbi += 1
                    inner = l["config"]["layer"]
                    extra = inner["config"].get("name") or "%s_%d" % (inner["class_name"].lower(), nbi)
                else:
                    extra = bool(l["config"].get("use_bias", True))
                specs.append((l["class_name"], l["name"], extra))
        else:
            cell = "gru" if self.config["GRU"] else "lstm"
            specs = [("InputLayer", "the_input", None), ("MaxPooling2D", "max_pooling2d_1", None), ("Conv2D", "conv2d_1", True),
                     ("MaxPooling2D", "max_pooling2d_2", None), ("Conv2D", "conv2d_2", True), ("Flatten", "flatten_1", None),
                     ("Dense", "dense_1", None), ("Activation", "activation_1", None), ("Dense", "dense_2", None),
                     ("BilinearInterpolation", "bilinear_interpolation_1", None), ("ZeroPadding2D", "zero_padding2d_1", None)]
            npool = 2
            for i in range(1, 8):
                specs += [("DepthwiseConv2D", "depthwise_conv2d_%d" % i, None), ("BatchNormalization", "batch_normalization_%d" % (2 * i - 1), None),
                          ("ReLU", "re_lu_%d" % (2 * i - 1), None), ("Conv2D", "conv2d_%d" % (i + 2), False),
                          ("BatchNormalization", "batch_normalization_%d" % (2 * i), None), ("ReLU", "re_lu_%d" % (2 * i), None)]
                if BLOCK_POOL[i - 1]:
                    npool += 1
                    specs.append(("MaxPooling2D", "max_pooling2d_%d" % npool, None))
                specs.append(("Dropout", "dropout_%d" % i, None))
            specs += [("Reshape", "reshape", None), ("Dense", "dense1", None), ("Dropout", "dropout_8", None),
                      ("Bidirectional", "bidirectional_1", cell + "_1"), ("Bidirectional", "bidirectional_2", cell + "_2"),
                      ("Dropout", "dropout_9", None), ("Dense", "dense2", None), ("Activation", "softmax", None)]
            if not self.predictor:
                specs += [("InputLayer", "the_labels", None), ("InputLayer", "input_length", None), ("InputLayer", "label_length", None),
                          ("Lambda", "ctc", None)]
        out = []
        for kind, name, inner in specs:
            if kind == "Conv2D":
                # the two localisation convolutions carry a bias, the pointwise ones do not (utils.py:46-53,249-251)
                w = take([name + "/kernel:0", name + "/bias:0"] if inner else [name + "/kernel:0"])
            elif kind == "DepthwiseConv2D":
                w = take([name + "/depthwise_kernel:0"])
            elif kind == "BatchNormalization":
                w = take([name + "/" + k + ":0" for k in ("gamma", "beta", "moving_mean", "moving_variance")])
            elif kind == "Dense":
                w = take([name + "/kernel:0", name + "/bias:0"])
            elif kind == "Bidirectional":
                w = take(["%s/%s_%s/%s:0" % (name, d, inner, k) for d in ("forward", "backward") for k in ("kernel", "recurrent_kernel", "bias")])
            else:
                w = []
            out.append((name, w))
        if next(widx, None) is not None:
            raise ValueError("Keras layer list does not cover every weight of the model")
        return out

    def _weights_group(self, group):
        """Fill `group` the way `save_weights_to_hdf5_group` does."""
        self._pull()
        st = self._state
        layers = self._keras_layers()
        group.attrs["layer_names"] = np.array([n.encode("utf8") for n, _ in layers])
        group.attrs["backend"] = b"tensorflow"
        group.attrs["keras_version"] = b"2.2.2"
        for lname, ws in layers:
            g = group.create_group(lname)
            g.attrs["weight_names"] = np.array([n.encode("utf8") for n, _ in ws]) if ws else np.zeros((0,), np.float64)
            for wname, (kind, key) in ws:
                g.create_dataset(wname, self._keras_shape(key, np.asarray(st["params"][key] if kind == "p" else st["bn"][key], np.float32)))
        return group

    @staticmethod
    def _attr_list(attrs, name):
        """Keras splits attributes above 64 KB into name0, name1, ... (`load_attributes_from_hdf5_group`)."""
        if name in attrs:
            vals = attrs[name]
        else:
            vals, k = [], 0
            while "%s%d" % (name, k) in attrs:
                vals.extend(attrs["%s%d" % (name, k)]); k += 1
        return [v.decode("utf8") if isinstance(v, bytes) else str(v) for v in np.asarray(vals).ravel().tolist()]

    def _load_hdf5(self, path):
        root = hdf5.read(path)
        if "layer_names" not in root.attrs and "layer_names0" not in root.attrs and "model_weights" in root:
            root = root["model_weights"]                          # a `model.save` file
        weights = []
        for lname in self._attr_list(root.attrs, "layer_names"):
            g = root[lname]
            weights += [g[w].value for w in self._attr_list(g.attrs, "weight_names")]
        self.set_weights(weights)

    def save_weights(self, path):
        if str(path).endswith(".npz"):
            with open(path, "wb") as f:
                np.savez(f, **{"w%03d" % i: w for i, w in enumerate(self.get_weights())})
        else:
            hdf5.write(path, self._weights_group(hdf5.Group()))

    def load_weights(self, path):
        if hdf5.is_hdf5(path):
            self._load_hdf5(path)
        else:
            with np.load(path) as z:
                self.set_weights([z["w%03d" % i] for i in range(len([k for k in z.files if k.startswith("w")]))])

    def save(self, path):
        """Keras `model.save`: `model_config` + `/model_weights` (the reference never resumes optimizer state: train.py
        re-compiles, so no `optimizer_weights` group is written)."""
        root = hdf5.Group()
        root.attrs["keras_version"] = b"2.2.2"
        root.attrs["backend"] = b"tensorflow"
        root.attrs["model_config"] = self.to_json().encode("utf8")
        self._weights_group(root.create_group("model_weights"))
        hdf5.write(path, root)

    def to_json(self):
        """Keras-2.2.2 functional-model JSON (keras_json.model_json); a model built from a Keras model.json hands that JSON back."""
        if self._keras_json is not None:
            return json.dumps(self._keras_json)
        from .keras_json import model_json
        return json.dumps(model_json(self.config, predictor=self.predictor))

    def count_params(self):
        trainable = sum(int(np.prod(d)) for _, _, d in self._state["layout"].values())
        return trainable, sum(v.size for v in self._state["bn"].values())

    def summary(self, print_fn=None):
        pr = print_fn or print
        pr("_" * 65)
        pr("%-34s %-20s %s" % ("Layer (type)", "Shape", "Param #"))
        pr("=" * 65)
        for n, (_, size, dims) in self._state["layout"].items():
            pr("%-34s %-20s %d" % (n, str(tuple(dims)), size))
        tr, nt = self.count_params()
        pr("=" * 65)
        pr("Total params: {:,}".format(tr + nt))
        pr("Trainable params: {:,}".format(tr))
        pr("Non-trainable params: {:,}".format(nt))
        pr("_" * 65)

    def get_layer(self, name):
        return self._layers[name]

    # ---- training / inference ---------------------------------------------------------------------------------
    def compile(self, loss=None, optimizer=None, **kwargs):
        """loss={'ctc': lambda y_true, y_pred: y_pred} in the reference (train.py:192): the model output already
        is the CTC cost, so the training loss is its batch mean -- that is what the engine differentiates."""
        self.optimizer = optimizer
        self._iterations = 0
        eng = self._state.get("engine")
        if eng is not None:
            eng.opt_state.clear()            # a new optimizer starts from zero moments (shared dict: cleared for every engine)

    def _snapshot(self, batch):
        """Readf re-yields the SAME arrays it keeps filling (utils.py:468,495-511): what a step needs is taken out of them before
        the generator is advanced again.  Host (NumPy) batches go straight into the engine's page-locked staging buffers and on to
        the device on a copy stream (Engine.stage: the snapshot, the float64 -> float32 conversion and the H->D transfer of batch
        k+1 all happen while step k runs); anything else (device tensors) is copied as before.  -> StagedBatch | (x, lab, il, ll)"""
        x, lab, il, ll = self._unpack(batch)
        if lab is not None and il is not None and ll is not None and not any(hasattr(a, "is_cuda") for a in (x, lab, il, ll)):
            return self._engine(len(x)).stage(x, lab, il, ll)
        cp = lambda a: None if a is None else (a.clone() if hasattr(a, "is_cuda") else np.array(a, copy=True))
        return (x.clone() if hasattr(x, "is_cuda") else np.asarray(x, dtype=np.float32).copy()), cp(lab), cp(il), cp(ll)

    @staticmethod
    def _unpack(batch):
        inputs = batch[0] if isinstance(batch, (tuple, list)) else batch
        if isinstance(inputs, dict):
            return inputs["the_input"], inputs.get("the_labels"), inputs.get("input_length"), inputs.get("label_length")
        return inputs, None, None, None

    def _dist(self):
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                return dist, dist.get_world_size()
        except Exception:
            pass
        return None, 1

    def _train_on_batch_async(self, x, labels=None, input_length=None, label_length=None, collective_sync_check=True):
        """Enqueue one train step; returns the batch-mean loss as a DEVICE scalar (no host synchronisation).
        x: the images, or a StagedBatch (then it carries labels and lengths).
        collective_sync_check (data parallel): agree over all ranks whether any replica is out of sync before the step
        (`_replicas_need_sync`: a 4-byte all-reduce + read-back); fit_generator passes False -- it synchronised the replicas itself
        and must not pay a host synchronisation per step."""
        if self.optimizer is None:
            raise RuntimeError("compile(optimizer=...) first")
        eng = self._engine(len(x))
        dist, world = self._dist()
        allreduce = None
        if world > 1:
            from .parallel import GradAllReduce
            # data parallel: a loop driven by train_on_batch must not run on per-rank random initial weights (or on weights only one
            # rank loaded) with averaged gradients -- the replicas would drift apart silently.  The out-of-sync mark is rank-local, so
            # the decision is taken by all ranks together: every rank broadcasts, or none does.
            if collective_sync_check and self._replicas_need_sync(eng, dist):
                self.sync_replicas(len(x))
            allreduce = GradAllReduce(eng, dist, world)
        eng.train_step(x, labels, input_length, label_length, self.optimizer, self._iterations, allreduce=allreduce)
        self._iterations += 1
        return eng.loss_and_status(), eng       # [batch-mean loss, give-up counter of the persistent recurrences]

    @staticmethod
    def _batch_len(b):
        return len(b[0]) if isinstance(b, tuple) else len(b)

    @staticmethod
    def _read_loss(ls_dev, eng, summed=False):
        """The one host synchronisation of a step: loss and recurrence status in one D2H copy; raises if a recurrence gave up.
        summed: the counter was all-reduced (SUM) over the data-parallel ranks -- compared against its own baseline."""
        loss, giveups = ls_dev.tolist()[:2]
        eng.raise_if_rnn_gave_up(giveups, summed=summed)
        return float(loss)

    def train_on_batch(self, x, labels, input_length, label_length):
        return self._read_loss(*self._train_on_batch_async(x, labels, input_length, label_length))

    def test_on_batch(self, x, labels, input_length, label_length):
        """learning_phase=0 forward + CTC cost (validation loss)."""
        import torch
        from .engine import _ptr, _stream
        eng = self._engine(len(x))
        y = eng.forward(x, train=False)
        eng._ctc_inputs(labels, input_length, label_length)      # host-side validation: the CTC kernel indexes LDS with the label ids
        scratch = eng.ws_tensor("dlogits")
        native.check(eng.lib.crnn_ctc_loss_grad(_ptr(y), _ptr(eng._lab), _ptr(eng._il), _ptr(eng._ll), _ptr(eng.loss), _ptr(scratch), eng.B, eng.T,
                                                eng.C, self.config["max_string_len"], 2, 0.0, _stream()), "ctc")
        loss, giveups = eng.loss_and_status().tolist()
        eng.raise_if_rnn_gave_up(giveups)
        return float(loss)

    def predict_on_batch(self, x):
        eng = self._engine(len(x))
        y = eng.forward(x, train=False).cpu().numpy()
        eng.check_rnn_status()
        return y

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, validation_data=None, validation_steps=None,
                      shuffle=False, verbose=1, callbacks=None, **kwargs):
        H = History()
        callbacks = list(callbacks or [])
        for cb in callbacks:
            cb.set_model(self)
            cb.on_train_begin({})
        self.stop_training = False
        dist, world = self._dist()
        if world > 1:
            # the documented collective point: replicas start from rank 0's weights (engine for the first batch's size built here)
            if self._prefetched is None or self._prefetched[0] is not generator:
                self._prefetched = (generator, self._snapshot(next(generator)))
            self.sync_replicas(self._batch_len(self._prefetched[1]))
        for epoch in range(epochs):
            t0 = time.time()
            run, nimg = 0.0, 0
            for step in range(steps_per_epoch):
                # the step is enqueued first, then the NEXT batch is drawn from the generator (host decode / augmentation)
                # while the GPU works, and only then is the loss read back (the one host sync per step Keras has too)
                if self._prefetched is None or self._prefetched[0] is not generator:   # a batch drawn ahead belongs to ITS generator
                    self._prefetched = (generator, self._snapshot(next(generator)))
                cur = self._prefetched[1]
                self._prefetched = None        # consumed: if drawing / staging the next batch raises, this one is not trained twice on a retry
                nb = self._batch_len(cur)
                ls_dev, eng = self._train_on_batch_async(*((cur,) if not isinstance(cur, tuple) else cur), collective_sync_check=False)
                pending = None
                try:
                    self._prefetched = (generator, self._snapshot(next(generator)))
                except Exception as e:         # (a bad batch k+1 / an exhausted generator): step k is already enqueued -- read it back and
                    pending = e                # run its callbacks first, then re-raise
                resync = False
                if world > 1:
                    # every rank logs (and EarlyStoppingIter monitors) the GLOBAL batch-mean loss, so all ranks take the same
                    # stop / restore decisions and keep issuing the same collectives; the give-up counters are summed, so a
                    # recurrence that gave up on one rank stops every rank (the rank-local counter keeps its own baseline).
                    # Third slot (round 6, ADVICE): this rank's out-of-sync mark -- a callback that called set_weights / load_weights on ONE
                    # rank since the last step -- rides on the same all-reduce and the same read-back, so the pipelined loop needs no
                    # extra collective or host synchronisation to notice it; every rank then takes the broadcast below together.
                    import torch
                    mark = torch.full((1,), 0.0 if self._state.get("replicas_synced") else 1.0, dtype=ls_dev.dtype, device=ls_dev.device)
                    ls_dev = torch.cat([ls_dev, mark])
                    dist.all_reduce(ls_dev, op=dist.ReduceOp.SUM)
                    ls_dev[0] /= world
                vals = ls_dev.tolist()                # the one host synchronisation of the step
                eng.raise_if_rnn_gave_up(vals[1], summed=world > 1)
                loss = float(vals[0])
                resync = world > 1 and vals[2] > 0
                run += loss; nimg += nb
                logs = {"loss": loss, "batch": step, "size": nb}
                for cb in callbacks:
                    cb.on_batch_end(step, logs)
                if resync:      # some rank's weights were replaced during the last step's callbacks: all ranks adopt rank 0's, one step late
                    self.sync_replicas(nb)
                if pending is not None:
                    raise pending
                if verbose and (step + 1 == steps_per_epoch or (step + 1) % max(1, steps_per_epoch // 20) == 0):
                    print("\r%d/%d - loss: %.4f - %.0f img/s" % (step + 1, steps_per_epoch, run / (step + 1), nimg / max(time.time() - t0, 1e-9)),
                          end="", flush=True)
                if self.stop_training:
                    break
            logs = {"loss": run / max(1, step + 1)}
            self._dp_sync_bn()                # moving statistics averaged over ranks before validation / checkpoints
            if validation_data is not None and validation_steps:
                logs["val_loss"] = self.evaluate_generator(validation_data, validation_steps)
            if verbose:
                print("\nEpoch %d/%d - %ds - %s" % (epoch + 1, epochs, time.time() - t0, " - ".join("%s: %.4f" % kv for kv in logs.items())))
            H.epoch.append(epoch)
            for k, v in logs.items():
                H.history.setdefault(k, []).append(v)
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end({})
        return H

    def _dp_sync_bn(self):
        dist, world = self._dist()
        eng = self._state.get("engine")
        if world > 1 and eng is not None:
            from .parallel import sync_bn_stats
            sync_bn_stats(eng, dist, world)

    def evaluate_generator(self, generator, steps):
        tot = 0.0
        for _ in range(steps):
            x, lab, il, ll = self._unpack(next(generator))
            tot += self.test_on_batch(x, lab, il, ll)
        return tot / max(1, steps)

    def predict_generator(self, generator, steps, **kwargs):
        """-> ndarray (steps*B, T, num_classes) float32 softmax (predict.py:166)."""
        outs = []
        for _ in range(steps):
            x, _, _, _ = self._unpack(next(generator))
            outs.append(self.predict_on_batch(x))
        return np.concatenate(outs, 0)


def init_predictor(model):
    """utils.py:308-312: the softmax sub-model sharing the trained weights."""
    return Model(model.config, predictor=True, share=model)


def crnn_config_from_keras_json(cfg):
    """Architecture facts of a Keras-2.2.2 `model.json` written by the reference (`save_model_json`, utils.py:530-533;
    e.g. models/OCR_mjsynth_FULL_2/model.json) -> the CRNN(...) constructor arguments.  Only the graph CRNN.get_model
    builds (utils.py:58-96) is accepted: anything else raises ValueError instead of silently building a different net."""
    layers = cfg["layers"]
    by_name = {l["name"]: l for l in layers}
    kinds = [l["class_name"] for l in layers]

    def need(cond, what):
        if not cond:
            raise ValueError("model.json is not the CRNN-OCR-lite graph: " + what)

    for n in ("the_input", "dense1", "dense2"):
        need(n in by_name, "layer %r missing" % n)
    shape = tuple(int(v) for v in by_name["the_input"]["config"]["batch_input_shape"][1:])
    need(len(shape) == 3 and shape[2] == 1, "the_input must be (H, W, 1)")
    need(kinds.count("DepthwiseConv2D") == len(BLOCK_FILTERS), "expected %d depthwise-separable blocks" % len(BLOCK_FILTERS))
    pw = [int(l["config"]["filters"]) for l in layers if l["class_name"] == "Conv2D" and list(l["config"].get("kernel_size", [])) == [1, 1]]
    need(pw == list(BLOCK_FILTERS), "pointwise filters %r != %r" % (pw, list(BLOCK_FILTERS)))
    birnn = [l for l in layers if l["class_name"] == "Bidirectional"]
    need(len(birnn) == 2, "expected two Bidirectional layers")
    cells = {l["config"]["layer"]["class_name"] for l in birnn}
    need(len(cells) == 1 and cells <= {"GRU", "LSTM"}, "recurrent cells %r" % sorted(cells))
    need([l["config"].get("merge_mode") for l in birnn] == ["sum", "concat"], "merge modes must be sum, concat (utils.py:77-82)")
    units = {int(l["config"]["layer"]["config"]["units"]) for l in birnn}
    need(len(units) == 1, "both recurrent layers must have the same width")
    for l in birnn:
        need(l["config"]["layer"]["config"].get("recurrent_activation", "hard_sigmoid") == "hard_sigmoid", "recurrent_activation")
    max_len = int(by_name["the_labels"]["config"]["batch_input_shape"][1]) if "the_labels" in by_name else 23
    return dict(num_classes=int(by_name["dense2"]["config"]["units"]), max_string_len=max_len, shape=shape,
                time_dense_size=int(by_name["dense1"]["config"]["units"]), GRU=(cells == {"GRU"}), n_units=units.pop())


def model_from_json(text, custom_objects=None):
    """Accepts this package's own `to_json` and the reference's Keras-2.2.2 `model.json` artefacts."""
    top = json.loads(text)
    cfg = top["config"]
    if "crnn" in cfg:
        crnn = cfg["crnn"]
        crnn["shape"] = tuple(crnn["shape"])
        return Model(crnn, predictor=cfg.get("predictor", False))
    if "layers" in cfg:
        crnn = crnn_config_from_keras_json(cfg)
        predictor = not any(l["name"] == "ctc" for l in cfg["layers"])      # init_predictor's sub-model has no loss head
        model = Model(crnn, predictor=predictor)
        model._keras_json = top
        return model
    raise ValueError("unrecognised model.json")


def save_model_json(model, save_path, model_name):
    """utils.py:530-533"""
    with open(save_path + '/' + model_name + "/model.json", "w") as f:
        f.write(model.to_json())


def load_custom_model(model_path, model_name='/model.json', weights="/final_weights.h5"):
    """utils.py:323-329"""
    with open(model_path + model_name, 'r') as f:
        model = model_from_json(f.read())
    model.load_weights(model_path + weights)
    return model


def load_model_custom(path, weights="model"):
    """utils.py:300-306"""
    return load_custom_model(path, '/model.json', "/%s.h5" % weights)


def ctc_lambda_func(args):
    """utils.py:98-103 on arrays: per-sample CTC cost of y_pred[:, 2:, :] (HIP kernel)."""
    import torch
    from .engine import _ptr, _stream
    y_pred, labels, input_length, label_length = args
    y = torch.from_numpy(np.ascontiguousarray(y_pred, dtype=np.float32)).cuda()
    B, T, C = y.shape
    lab = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).cuda()
    il = torch.from_numpy(np.ascontiguousarray(np.asarray(input_length).reshape(-1), dtype=np.int32)).cuda()
    ll = torch.from_numpy(np.ascontiguousarray(np.asarray(label_length).reshape(-1), dtype=np.int32)).cuda()
    loss = torch.empty(B, dtype=torch.float32, device="cuda")
    scratch = torch.empty(T * B * C, dtype=torch.float32, device="cuda")
    native.check(native.lib().crnn_ctc_loss_grad(_ptr(y), _ptr(lab), _ptr(il), _ptr(ll), _ptr(loss), _ptr(scratch), B, T, C, lab.shape[1], 2, 0.0,
                                                 _stream()), "ctc")
    return loss.cpu().numpy().reshape(B, 1)


class BilinearInterpolation:
    """utils.py:116-237 as a callable on arrays: layer([image (B,H,W,1), theta (B,6)]) -> (B,H,W,1)."""

    def __init__(self, output_size=(100, 32), **kwargs):
        self.output_size = output_size

    def compute_output_shape(self, input_shapes):
        return (None, self.output_size[0], self.output_size[1], input_shapes[0][-1])

    def get_config(self):
        return {"output_size": self.output_size}

    def __call__(self, tensors, mask=None):
        import torch
        from .engine import _ptr, _stream
        X, theta = tensors
        X = np.asarray(X, dtype=np.float32)
        B, H, W, Cc = X.shape
        if Cc != 1 or (H, W) != tuple(self.output_size):
            raise native.CrnnError("the HIP sampler handles single-channel maps with output_size == input size")
        xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()
        td = torch.from_numpy(np.ascontiguousarray(np.asarray(theta, dtype=np.float32).reshape(B, 6))).cuda()
        out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
        native.check(native.lib().crnn_sampler_fwd(_ptr(xd), _ptr(td), _ptr(out), B, H, W, 0, _stream()), "sampler")
        return out.cpu().numpy()[..., None]

    call = __call__


def get_initial_weights(output_size):
    """utils.py:239-245"""
    b = np.zeros((2, 3), dtype='float32')
    b[0, 0] = 1
    b[1, 1] = 1
    return [np.zeros((output_size, 6), dtype='float32'), b.flatten()]


def STN(image, sampling_size=(100, 32), weights=None):
    """utils.py:247-258 on arrays: MaxPool(2,2) -> Conv2D(20,5x5) -> MaxPool(2,2) -> Conv2D(20,5x5) -> Flatten -> Dense(50) ->
    relu -> Dense(6, weights=get_initial_weights(50)) -> BilinearInterpolation(sampling_size), each stage a HIP kernel of the
    fused model's spatial transformer (csrc/stn.hip) called through the C ABI.
    image (B,H,W,1) ndarray -> (B,H,W,1) float32.  weights: the 8 arrays of the localisation net in Keras order
    [conv2d_1 kernel (5,5,1,20), bias, conv2d_2 kernel (5,5,20,20), bias, dense_1 kernel (F,50), bias, dense_2 kernel (50,6), bias];
    None = what the reference's freshly built layers hold: glorot-uniform convs / dense_1, zero biases and the identity transform
    in dense_2 (get_initial_weights) -- with dense_2's kernel all zero theta is the identity whatever the other layers hold."""
    import torch
    from .engine import _ptr, _stream
    X = np.ascontiguousarray(np.asarray(image, dtype=np.float32))
    if X.ndim != 4 or X.shape[3] != 1:
        raise ValueError("STN expects (B, H, W, 1) images, got %r" % (X.shape,))
    B, H, W, _ = X.shape
    if tuple(sampling_size) != (H, W):
        raise native.CrnnError("the HIP sampler handles output_size == input size (the reference always calls STN(inputs, shape[:2]))")
    Hs1, Ws1 = H // 2, W // 2
    Ho1, Wo1 = Hs1 - 4, Ws1 - 4
    Hs2, Ws2 = Ho1 // 2, Wo1 // 2
    Ho2, Wo2 = Hs2 - 4, Ws2 - 4
    if Ho2 < 1 or Wo2 < 1:
        raise ValueError("image too small for the localisation net (two 2x2 pools and two 5x5 valid convs)")
    F = Ho2 * Wo2 * 20
    if weights is None:
        rs = np.random.RandomState()
        glorot = lambda shape, fi, fo: rs.uniform(-np.sqrt(6.0 / (fi + fo)), np.sqrt(6.0 / (fi + fo)), shape).astype(np.float32)
        weights = [glorot((5, 5, 1, 20), 25, 500), np.zeros(20, np.float32), glorot((5, 5, 20, 20), 500, 500), np.zeros(20, np.float32),
                   glorot((F, 50), F, 50), np.zeros(50, np.float32)] + get_initial_weights(50)
    shapes = [(5, 5, 1, 20), (20,), (5, 5, 20, 20), (20,), (F, 50), (50,), (50, 6), (6,)]
    if len(weights) != 8:
        raise ValueError("expected the 8 weight arrays of the localisation net, got %d" % len(weights))
    dev = []
    for w, shp in zip(weights, shapes):
        w = np.asarray(w, dtype=np.float32)
        if w.shape != shp:
            raise ValueError("localisation-net weight of shape %r, expected %r" % (w.shape, shp))
        dev.append(torch.from_numpy(np.ascontiguousarray(w)).cuda())
    k1, b1, k2, b2, w1, c1b, w2, c2b = dev
    L = native.lib()
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device="cuda")
    xd = torch.from_numpy(X).cuda()
    pool1, c1, pool2, flat = f32(B, Hs1, Ws1), f32(B, Ho1, Wo1, 20), f32(B, Hs2, Ws2, 20), f32(B, F)
    fc1, theta, out = f32(B, 50), f32(B, 6), f32(B, H, W)
    st = _stream()
    native.check(L.crnn_maxpool_fwd(_ptr(xd), _ptr(pool1), B, H, W, 1, 2, 2, st), "maxpool")
    native.check(L.crnn_loc_conv_fwd(_ptr(pool1), _ptr(k1), _ptr(b1), _ptr(c1), B, Hs1, Ws1, 1, st), "loc_conv 1")
    native.check(L.crnn_maxpool_fwd(_ptr(c1), _ptr(pool2), B, Ho1, Wo1, 20, 2, 2, st), "maxpool")
    native.check(L.crnn_loc_conv_fwd(_ptr(pool2), _ptr(k2), _ptr(b2), _ptr(flat), B, Hs2, Ws2, 20, st), "loc_conv 2")
    native.check(L.crnn_loc_fc_fwd(_ptr(flat), _ptr(w1), _ptr(c1b), _ptr(w2), _ptr(c2b), _ptr(fc1), _ptr(theta), B, F, st), "loc_fc")
    native.check(L.crnn_sampler_fwd(_ptr(xd), _ptr(theta), _ptr(out), B, H, W, 0, st), "sampler")
    return out.cpu().numpy()[..., None]
