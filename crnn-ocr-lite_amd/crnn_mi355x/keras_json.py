"""Keras-2.2.2 functional-model JSON of the graph CRNN.get_model() builds (utils.py:58-96, STN utils.py:247-258), so that the
`model.json` written by train.py / `save_model_json` and the `model_config` attribute of `model.save` files have the schema Keras'
`model_from_json` reads (class_name / config / inbound_nodes per layer, input_layers, output_layers) -- the reverse direction of
surface.crnn_config_from_keras_json.  Layer names are those a fresh Keras session assigns (the ones in the reference's
models/*/model.json).

One thing cannot be reproduced: Keras stores the `ctc` Lambda as marshalled CPython-3.6 bytecode.  It is written here by NAME
("function_type": "function", "function": "ctc_lambda_func"); Keras resolves that through `custom_objects={'ctc_lambda_func': ...}`
(the reference's load_custom_model passes only BilinearInterpolation, so loading the *training* graph there needs that one extra
entry; the predictor graph has no Lambda and loads as is)."""

BLOCK_FILTERS = (64, 128, 256, 256, 512, 512, 512)
BLOCK_POOL = (None, None, (2, 2), None, (1, 2), None, None)


def _init(kind, **cfg):
    return {"class_name": kind, "config": cfg}


_GLOROT = _init("VarianceScaling", scale=1.0, mode="fan_avg", distribution="uniform", seed=None)
_HE_NORMAL = _init("VarianceScaling", scale=2.0, mode="fan_in", distribution="normal", seed=None)
_ZEROS, _ONES = _init("Zeros"), _init("Ones")
_NO_REG = {"kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}


def model_json(config, predictor=False):
    """config: the CRNN constructor arguments (num_classes, max_string_len, shape, time_dense_size, GRU, n_units)."""
    H, W, Cin = (int(v) for v in config["shape"])
    units, tds, ncls, max_len = int(config["n_units"]), int(config["time_dense_size"]), int(config["num_classes"]), int(config["max_string_len"])
    layers = []
    prev = [None]

    def add(kind, name, cfg, inbound=None, track=True):
        full = {"name": name}
        if kind != "InputLayer":
            full["trainable"] = True
        full.update(cfg)
        if inbound is None:
            inbound = [prev[0]] if prev[0] else []
        nodes = [[[n, 0, 0, {}] for n in inbound]] if inbound else []
        layers.append({"name": name, "class_name": kind, "config": full, "inbound_nodes": nodes})
        if track:
            prev[0] = name
        return name

    def conv(name, filters, ksize, padding, use_bias):
        cfg = {"filters": filters, "kernel_size": list(ksize), "strides": [1, 1], "padding": padding, "data_format": "channels_last",
               "dilation_rate": [1, 1], "activation": "linear", "use_bias": use_bias, "kernel_initializer": _GLOROT, "bias_initializer": _ZEROS}
        cfg.update(_NO_REG)
        return add("Conv2D", name, cfg)

    def dense(name, n, activation, kernel_init=_GLOROT, bias_init=_ZEROS):
        cfg = {"units": n, "activation": activation, "use_bias": True, "kernel_initializer": kernel_init, "bias_initializer": bias_init}
        cfg.update(_NO_REG)
        return add("Dense", name, cfg)

    def pool(name, size):
        return add("MaxPooling2D", name, {"pool_size": list(size), "padding": "valid", "strides": list(size), "data_format": "channels_last"})

    add("InputLayer", "the_input", {"batch_input_shape": [None, H, W, Cin], "dtype": "float32", "sparse": False})
    # spatial transformer (utils.py:247-258)
    pool("max_pooling2d_1", (2, 2)); conv("conv2d_1", 20, (5, 5), "valid", True)
    pool("max_pooling2d_2", (2, 2)); conv("conv2d_2", 20, (5, 5), "valid", True)
    add("Flatten", "flatten_1", {"data_format": "channels_last"})
    dense("dense_1", 50, "linear")
    add("Activation", "activation_1", {"activation": "relu"})
    dense("dense_2", 6, "linear")
    add("BilinearInterpolation", "bilinear_interpolation_1", {"output_size": [H, W]}, inbound=["the_input", "dense_2"])
    add("ZeroPadding2D", "zero_padding2d_1", {"padding": [[2, 2], [2, 2]], "data_format": "channels_last"})
    # 7 depthwise-separable blocks (utils.py:43-56,64-70)
    npool = 2
    hh, ww = H + 4, W + 4
    for i, (filters, pl) in enumerate(zip(BLOCK_FILTERS, BLOCK_POOL), 1):
        dcfg = {"kernel_size": [3, 3], "strides": [1, 1], "padding": "same", "data_format": "channels_last", "activation": "linear",
                "dilation_rate": [1, 1], "use_bias": False, "bias_initializer": _ZEROS, "bias_regularizer": None, "activity_regularizer": None, "bias_constraint": None,
                "depth_multiplier": 1, "depthwise_initializer": _GLOROT, "depthwise_regularizer": None, "depthwise_constraint": None}
        add("DepthwiseConv2D", "depthwise_conv2d_%d" % i, dcfg)
        for j, after in ((2 * i - 1, None), (2 * i, filters)):
            if after is not None:
                conv("conv2d_%d" % (i + 2), filters, (1, 1), "same", False)
            add("BatchNormalization", "batch_normalization_%d" % j,
                {"axis": -1, "momentum": 0.99, "epsilon": 0.001, "center": True, "scale": True, "beta_initializer": _ZEROS, "gamma_initializer": _ONES,
                 "moving_mean_initializer": _ZEROS, "moving_variance_initializer": _ONES, "beta_regularizer": None, "gamma_regularizer": None,
                 "beta_constraint": None, "gamma_constraint": None})
            add("ReLU", "re_lu_%d" % j, {"max_value": 6.0})
        if pl:
            npool += 1
            pool("max_pooling2d_%d" % npool, pl)
            hh, ww = hh // pl[0], ww // pl[1]
        add("Dropout", "dropout_%d" % i, {"rate": 0.1, "noise_shape": None, "seed": None})
    add("Reshape", "reshape", {"target_shape": [hh, ww * BLOCK_FILTERS[-1]]})
    dense("dense1", tds, "relu")
    add("Dropout", "dropout_8", {"rate": 0.4, "noise_shape": None, "seed": None})
    cell = "GRU" if config["GRU"] else "LSTM"
    for n, merge in ((1, "sum"), (2, "concat")):
        inner = {"name": "%s_%d" % (cell.lower(), n), "trainable": True, "return_sequences": True, "return_state": False, "go_backwards": False,
                 "stateful": False, "unroll": False, "units": units, "activation": "tanh", "recurrent_activation": "hard_sigmoid", "use_bias": True,
                 "kernel_initializer": _HE_NORMAL, "recurrent_initializer": _init("Orthogonal", gain=1.0, seed=None), "bias_initializer": _ZEROS,
                 "kernel_regularizer": None, "recurrent_regularizer": None, "bias_regularizer": None, "activity_regularizer": None,
                 "kernel_constraint": None, "recurrent_constraint": None, "bias_constraint": None, "dropout": 0.0, "recurrent_dropout": 0.0,
                 "implementation": 1}
        if cell == "GRU":
            inner["reset_after"] = False
        else:
            inner["unit_forget_bias"] = True
        add("Bidirectional", "bidirectional_%d" % n, {"layer": {"class_name": cell, "config": inner}, "merge_mode": merge})
    add("Dropout", "dropout_9", {"rate": 0.2, "noise_shape": None, "seed": None})
    dense("dense2", ncls, "linear", kernel_init=_HE_NORMAL)
    add("Activation", "softmax", {"activation": "softmax"})
    inputs, outputs = [["the_input", 0, 0]], [["softmax", 0, 0]]
    if not predictor:
        for name, shape in (("the_labels", [None, max_len]), ("input_length", [None, 1]), ("label_length", [None, 1])):
            add("InputLayer", name, {"batch_input_shape": shape, "dtype": "float32" if name == "the_labels" else "int64", "sparse": False}, inbound=[], track=False)
            inputs.append([name, 0, 0])
        add("Lambda", "ctc", {"function": "ctc_lambda_func", "function_type": "function", "output_shape": [1], "output_shape_type": "raw", "arguments": {}},
            inbound=["softmax", "the_labels", "input_length", "label_length"])
        outputs = [["ctc", 0, 0]]
    return {"class_name": "Model", "config": {"name": "model_1", "layers": layers, "input_layers": inputs, "output_layers": outputs},
            "keras_version": "2.2.2", "backend": "tensorflow"}
