"""Drop-in for the reference's `utils` module: `from utils import *` (train.py:119) and
`from utils import init_predictor, DecodeCTCPred, Readf, ...` (predict.py:101-102) resolve to the
MI355X-native implementations.  Put this directory (crnn-ocr-lite_amd/) on sys.path in place of the
reference checkout."""
import numpy as np  # noqa: F401  (the reference's star-import also leaks np / os / re / pickle)
import os, re, gc, pickle, math, string  # noqa: F401,E401
from numpy.random import RandomState  # noqa: F401

from crnn_mi355x import optimizers  # noqa: F401  (utils.py:28 re-exports keras.optimizers)
from crnn_mi355x.surface import (CRNN, Model, init_predictor, load_custom_model, load_model_custom, save_model_json,  # noqa: F401
                                 model_from_json, ctc_lambda_func, BilinearInterpolation, STN, get_initial_weights)
from crnn_mi355x.decode import DecodeCTCPred, labels_to_text  # noqa: F401
from crnn_mi355x.data import (Readf, open_img, read_img, norm, parse_mjsynth, get_lengths, get_lexicon, make_ohe)  # noqa: F401
from crnn_mi355x.metrics import levenshtein, edit_distance, normalized_edit_distance  # noqa: F401
from crnn_mi355x.callbacks import Callback, EarlyStoppingIter, ModelCheckpoint  # noqa: F401
