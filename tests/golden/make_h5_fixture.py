#!/usr/bin/env python
"""Generates tests/golden/h5py_*.h5 + h5py_expected.npz with h5py / libhdf5 (an independent HDF5 implementation).

h5py is not importable by the image's main interpreter; an unrelated conda environment has it:
    /opt/conda/bin/python3.9 -E -s tests/golden/make_h5_fixture.py
(h5py 3.3.0, HDF5 1.10.6).  The files are laid out the way Keras 2.2.2 `save_weights` / `model.save` lay theirs out
(keras/engine/saving.py: `layer_names`, `weight_names`, one group per layer, weight names containing '/'), plus the
HDF5 features a weight file from another Keras/h5py version may carry: variable-length string attributes, chunked +
deflate + shuffle datasets, more links than one symbol-table node holds, a two-level group B-tree, and the
`libver='latest'` object model (superblock v3, version-2 object headers, compact link messages).
`h5py_expected.npz` holds what h5py itself reads back from the files, keyed "<file>|d|<path>" / "<file>|a|<path>@<attr>".
"""
import os
import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rs = np.random.RandomState(7)
expected = {}


def keras_weights(g, layers):
    g.attrs["layer_names"] = [n.encode("utf8") for n, _ in layers]
    g.attrs["backend"] = "tensorflow".encode("utf8")
    g.attrs["keras_version"] = "2.2.2".encode("utf8")
    for name, weights in layers:
        lg = g.create_group(name)
        lg.attrs["weight_names"] = [w.encode("utf8") for w, _ in weights]
        for wname, shape in weights:
            val = rs.normal(size=shape).astype(np.float32)
            d = lg.create_dataset(wname, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


LAYERS = [("the_input", []), ("conv2d_1", [("conv2d_1/kernel:0", (5, 5, 1, 20)), ("conv2d_1/bias:0", (20,))]),
          ("flatten_1", []), ("dense_1", [("dense_1/kernel:0", (76, 50)), ("dense_1/bias:0", (50,))]),
          ("depthwise_conv2d_1", [("depthwise_conv2d_1/depthwise_kernel:0", (3, 3, 1, 1))]),
          ("batch_normalization_1", [("batch_normalization_1/" + k + ":0", (16,)) for k in ("gamma", "beta", "moving_mean", "moving_variance")]),
          ("re_lu_1", []), ("conv2d_3", [("conv2d_3/kernel:0", (1, 1, 16, 32))]), ("dropout_1", []), ("reshape", []),
          ("bidirectional_1", [("bidirectional_1/%s_lstm_1/%s:0" % (d, k), s) for d in ("forward", "backward")
                               for k, s in (("kernel", (8, 16)), ("recurrent_kernel", (4, 16)), ("bias", (16,)))]),
          ("dense2", [("dense2/kernel:0", (8, 5)), ("dense2/bias:0", (5,))]), ("softmax", []), ("ctc", [])]


def record(tag, f):
    def visit(path, node):
        for k, v in node.attrs.items():
            a = np.asarray(v if not isinstance(v, str) else v.encode("utf8"))
            if a.dtype.kind == "O":                            # variable-length strings: keep the UTF-8 bytes
                a = np.array([x.encode("utf8") if isinstance(x, str) else x for x in a.ravel()], dtype="S").reshape(a.shape)
            expected["%s|a|%s@%s" % (tag, path, k)] = a
        if isinstance(node, h5py.Dataset):
            v = node[()]
            expected["%s|d|%s" % (tag, path)] = np.asarray(v if not isinstance(v, (str, bytes)) else (v.encode("utf8") if isinstance(v, str) else v))
        else:
            expected["%s|g|%s" % (tag, path)] = np.array(sorted(node.keys()), dtype="S")
    visit("/", f)
    f.visititems(lambda p, n: visit("/" + p, n))


# 1. Keras save_weights layout, library defaults (the original object model)
with h5py.File(os.path.join(HERE, "h5py_keras_weights.h5"), "w") as f:
    keras_weights(f, LAYERS)
with h5py.File(os.path.join(HERE, "h5py_keras_weights.h5"), "r") as f:
    record("keras_weights", f)

# 2. Keras model.save layout + assorted HDF5 features
with h5py.File(os.path.join(HERE, "h5py_features.h5"), "w") as f:
    f.attrs["model_config"] = '{"class_name": "Model", "note": "café"}'          # str -> variable-length UTF-8
    f.attrs["keras_version"] = "2.2.2".encode("utf8")
    f.attrs["ints"] = np.arange(5, dtype=np.int32)
    f.attrs["scalar_f64"] = 2.5
    f.attrs["vlen_list"] = np.array(["ab", "", "longer string"], dtype=h5py.string_dtype())
    keras_weights(f.create_group("model_weights"), LAYERS[:6])
    m = f.create_group("misc")
    m.create_dataset("chunked_gzip", data=rs.normal(size=(37, 21)).astype(np.float32), chunks=(8, 8), compression="gzip", shuffle=True)
    m.create_dataset("chunked_plain", data=np.arange(100, dtype=np.int64).reshape(10, 10), chunks=(4, 3))
    m.create_dataset("fletcher", data=rs.normal(size=(9,)), chunks=(4,), fletcher32=True)
    m.create_dataset("scalar", data=np.float32(1.25))
    m.create_dataset("u8", data=np.arange(7, dtype=np.uint8))
    m.create_dataset("big_endian", data=np.arange(6, dtype=">f4"))
    m.create_dataset("f16", data=np.arange(6, dtype=np.float16) / 4)
    m.create_dataset("empty", shape=(0, 3), dtype=np.float32)
    m.create_dataset("unwritten", shape=(4,), dtype=np.float32)
    m.create_dataset("bytes", data=np.array([b"abc", b"de"], dtype="S3"))
    m.create_group("empty_group")
    many = f.create_group("many")                              # 300 links: 38 symbol-table nodes under a two-level B-tree
    for i in range(300):
        many.create_dataset("w_%03d:0" % i, data=np.float32(i))
with h5py.File(os.path.join(HERE, "h5py_features.h5"), "r") as f:
    record("features", f)

# 3. the same Keras layout written with the newest object model
with h5py.File(os.path.join(HERE, "h5py_latest.h5"), "w", libver="latest") as f:
    keras_weights(f, LAYERS[:6])
    f.create_dataset("compact_links/x", data=np.arange(4, dtype=np.float32))
with h5py.File(os.path.join(HERE, "h5py_latest.h5"), "r") as f:
    record("latest", f)

np.savez_compressed(os.path.join(HERE, "h5py_expected.npz"), **expected)
print("wrote", len(expected), "expected entries with h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)
