#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference's own code (/root/reference/utils.py).

Run here (build container) only:  python tests/golden/make_golden.py
The reference cannot travel to the GPU box, so the outputs (data only) are committed as
tests/golden/*.npz / *.json.  Keras / TensorFlow / cv2 are absent from this image, so the
module is imported over *stubs*; only code that the reference itself implements is executed:

  * BilinearInterpolation._transform/_interpolate/_make_regular_grids (utils.py:140-232) run
    over a NumPy shim of the ~15 `K.*` / `tf.*` backend functions they call (shim semantics:
    K.cast(float->int32) truncates toward zero like tf.cast; K.gather = take rows;
    K.batch_dot = batched matmul);
  * pure-Python helpers: levenshtein / edit_distance / normalized_edit_distance (262-298),
    get_lexicon (524-528), Readf.make_target / get_blank_matrices (435-452), norm (415-416),
    DecodeCTCPred.labels_to_text (338-345), labels_to_text (314-321), get_initial_weights
    (239-245), parse_mjsynth (412-413), EarlyStoppingIter.on_batch_end logic (587-610).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/utils.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

    mod("cv2")
    layer_names = ("Conv2D MaxPooling2D Activation Dropout add Dense Input Lambda Bidirectional "
                   "ZeroPadding2D concatenate Flatten multiply ReLU DepthwiseConv2D TimeDistributed "
                   "MaxPool2D LSTM GRU").split()
    keras = mod("keras")
    layers = mod("keras.layers", **{n: _Any for n in layer_names})
    core = mod("keras.layers.core", Reshape=_Any)
    core.__all__ = ["Reshape"]
    mod("keras.layers.normalization", BatchNormalization=_Any)

    class Callback:
        def __init__(self):
            self.model = None

    mod("keras.callbacks", Callback=Callback)
    mod("keras.models", Model=_Any, load_model=_Any, model_from_json=_Any)
    mod("keras.optimizers")
    keras.optimizers = sys.modules["keras.optimizers"]

    class Layer:
        def __init__(self, **kwargs):
            pass

    mod("keras.engine")
    mod("keras.engine.topology", Layer=Layer)

    # ---- NumPy shim of the Keras backend functions used by utils.py:140-232
    K = mod("keras.backend")
    K.shape = lambda x: x.shape
    K.int_shape = lambda x: x.shape
    K.cast = lambda x, dtype: (np.asarray(x).astype(dtype) if not np.isscalar(x) else np.dtype(dtype).type(x))
    K.flatten = lambda x: np.asarray(x).reshape(-1)
    K.clip = lambda x, lo, hi: np.clip(x, lo, hi)
    K.arange = lambda a, b: np.arange(a, b)
    K.expand_dims = lambda x, axis=-1: np.expand_dims(x, axis)
    K.repeat_elements = lambda x, rep, axis: np.repeat(x, rep, axis=axis)
    K.reshape = lambda x, shape: np.reshape(x, shape)
    K.gather = lambda ref, idx: ref[idx]
    K.ones_like = lambda x: np.ones_like(x)
    K.concatenate = lambda xs, axis=-1: np.concatenate(xs, axis=axis)
    K.tile = lambda x, n: np.tile(x, n)
    K.stack = lambda xs: np.stack(xs)
    K.batch_dot = lambda a, b: np.matmul(a, b)
    keras.backend = K
    tf = mod("tensorflow")
    tf.meshgrid = lambda x, y: np.meshgrid(x, y)
    tf.linspace = lambda a, b, n: np.linspace(a, b, n).astype(np.float32)


def main():
    _install_stubs()
    if not hasattr(np, "Inf"):
        np.Inf = np.inf  # the reference targets NumPy 1.x (utils.py:585)
    spec = importlib.util.spec_from_file_location("ref_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rs = np.random.RandomState(1234)
    # ---------------- sampler golden vectors (float32, as the reference computes) ---------
    cases = {}
    for name, (B, H, W, C) in {"mj": (3, 100, 32, 1), "iam": (2, 200, 32, 1), "small": (2, 12, 10, 3)}.items():
        img = rs.uniform(-3, 3, size=(B, H, W, C)).astype(np.float32)
        ident = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, 1))
        pert = (ident + rs.uniform(-0.25, 0.25, size=(B, 6))).astype(np.float32)
        wild = (ident * 1.6 + rs.uniform(-0.6, 0.6, size=(B, 6))).astype(np.float32)  # leaves the image
        layer = ref.BilinearInterpolation(output_size=(H, W))
        for tn, th in (("ident", ident), ("pert", pert), ("wild", wild)):
            out = layer._transform(img, th, (H, W))
            cases[f"{name}_{tn}_img"] = img
            cases[f"{name}_{tn}_theta"] = th
            cases[f"{name}_{tn}_out"] = np.asarray(out, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "sampler_golden.npz"), **cases)

    # ---------------- helper known-answers ------------------------------------------------
    lex = ref.get_lexicon()
    classes = {j: i for i, j in enumerate(lex)}
    inv = {v: k for k, v in classes.items()}
    reader = ref.Readf(img_size=(100, 32, 1), max_len=23, normed=True, batch_size=4, classes=classes)
    X, Y, il, ll = reader.get_blank_matrices()
    words = ["hello-w0rld!", "cellist", "overfilled", "beginning", "a", "Zz9_-"]
    pairs = [("overfiled", "overfilled"), ("celist", "cellist"), ("", "abc"), ("kitten", "sitting"),
             ("begining", "beginning"), ("same", "same")]
    dec = ref.DecodeCTCPred(top_paths=1, beam_width=10, inverse_classes=inv)
    lab_cases = [[17, 14, 21, 21, 24, 37, -1, -1], [37, 37], [0, 9, 10, 35, 36, -1], []]
    W0, b0 = ref.get_initial_weights(50)

    # EarlyStoppingIter trace (utils.py:587-610)
    class _M:
        def __init__(self):
            self.stop_training = False
            self.w = 0

        def get_weights(self):
            return self.w

        def set_weights(self, w):
            self.w = w

    es = ref.EarlyStoppingIter(monitor="loss", min_delta=.0001, patience=3, restore_best_weights=True, mode="auto")
    es.model = _M()
    es.on_train_begin()
    losses = [5.0, 4.0, 3.0, 2.5, 2.4, 2.6, 3.5, 4.5, 5.5, 6.5, 9.0, 9.0, 9.0]
    trace = []
    for i, l in enumerate(losses):
        es.model.w = i
        es.on_batch_end(i, {"loss": l})
        trace.append([bool(es.model.stop_training), float(es.best), int(es.stopped_iter)])
        if es.model.stop_training:
            break

    golden = {
        "lexicon": lex,
        "make_target": {w: [int(v) for v in reader.make_target(w.lower() if w != "Zz9_-" else w)] for w in words},
        "blank_matrices": {"X_shape": list(X.shape), "X_dtype": str(X.dtype), "Y": Y.tolist(), "Y_dtype": str(Y.dtype),
                           "input_length": il.tolist(), "label_length": ll.tolist()},
        "norm_in": [0, 118, 255, 37],
        "norm_out": [float(v) for v in ref.norm(np.array([0, 118, 255, 37]), reader.mean, reader.std)],
        "norm_dtype": str(ref.norm(np.array([0, 118, 255, 37]), reader.mean, reader.std).dtype),
        "levenshtein": [[a, b, float(ref.levenshtein(a, b))] for a, b in pairs],
        "edit_distance": float(ref.edit_distance([a for a, _ in pairs if _], [b for a, b in pairs if b])),
        "normalized_edit_distance": float(ref.normalized_edit_distance([a for a, b in pairs if b], [b for a, b in pairs if b])),
        "labels_to_text": [[l, dec.labels_to_text(l)] for l in lab_cases],
        "labels_to_text_fn": [[l, ref.labels_to_text(l, inv)] for l in lab_cases],
        "initial_weights": {"W_shape": list(W0.shape), "W_absmax": float(np.abs(W0).max()), "b": b0.tolist()},
        "parse_mjsynth": ref.parse_mjsynth("/data/mj", ["./2425/1/115_Lube_45484.jpg 45484\n", "./1/2/3_a_1.jpg 1"]),
        "early_stopping": {"losses": losses, "patience": 3, "min_delta": 0.0001, "trace": trace,
                           "final_weights": es.model.w},
        # known-answer beam-decode pairs pinned by the reference's screenshots:
        # TF-1.8 merge_repeated=True collapses adjacent duplicates of the decoded string.
        "beam_known_answers": [["cellist", "celist"], ["overfilled", "overfiled"], ["beginning", "begining"]],
    }
    with open(os.path.join(OUT, "helpers_golden.json"), "w") as f:
        json.dump(golden, f, indent=1)

    # ---- the Keras-2.2.2 model.json artefacts the reference ships (models/*/model.json): keep the architecture
    # facts a loader needs (layer class / name / units / shapes; data, not code) + what its arguments.txt and
    # model_summary.txt say, as the known answers for crnn_mi355x.surface.model_from_json
    keep = ("batch_input_shape", "units", "filters", "kernel_size", "pool_size", "padding", "rate", "merge_mode", "activation",
            "axis", "momentum", "epsilon", "output_size", "depth_multiplier", "use_bias", "max_value")
    arts = {}
    for name in sorted(os.listdir(os.path.join(os.path.dirname(REF), "models"))):
        mj = json.load(open(os.path.join(os.path.dirname(REF), "models", name, "model.json")))
        layers = []
        for l in mj["config"]["layers"]:
            cfg = {k: l["config"][k] for k in keep if k in l["config"]}
            if l["class_name"] == "Bidirectional":
                inner = l["config"]["layer"]
                cfg["layer"] = {"class_name": inner["class_name"],
                                "config": {k: inner["config"][k] for k in ("units", "activation", "recurrent_activation", "return_sequences",
                                                                           "implementation", "reset_after") if k in inner["config"]}}
            layers.append({"class_name": l["class_name"], "name": l["name"], "config": cfg})
        summary = open(os.path.join(os.path.dirname(REF), "models", name, "model_summary.txt")).read()
        counts = {k: int(v.replace(",", "")) for k, v in
                  __import__("re").findall(r"(Total params|Trainable params|Non-trainable params): ([0-9,]+)", summary)}
        arts[name] = {"model_json": {"class_name": mj["class_name"], "keras_version": mj["keras_version"], "backend": mj["backend"],
                                     "config": {"name": mj["config"].get("name"), "layers": layers}},
                      "param_counts": counts}
    with open(os.path.join(OUT, "keras_model_json.json"), "w") as f:
        json.dump(arts, f, indent=0)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
