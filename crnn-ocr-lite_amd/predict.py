#!/usr/bin/env python3
"""Prediction CLI with the reference's flag surface (predict.py:62-79): forward on the GPU, whole-batch
HIP beam-search decode (beam_width=10, top_paths=1, merge_repeated as TF 1.8), optional edit-distance report
and prediction.csv."""
import argparse
import os
import pickle
import re
import time

import numpy as np
from numpy.random import RandomState


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--model_path', type=str, required=True)
    parser.add_argument('--image_path', type=str, required=True)
    parser.add_argument('--result_path', type=str, required=False, default=None)
    parser.add_argument('--max_len', type=int, required=False, default=23)
    parser.add_argument('--boxes', type=str, required=False, default=None)
    parser.add_argument('--val_fname', type=str, required=False, default=None)
    parser.add_argument('--num_instances', type=int, default=None)
    parser.add_argument('--G', type=int, default=-1)
    parser.add_argument('--batch_size', type=int, default=64)
    parser.add_argument('--random_state', type=int, default=42)
    parser.add_argument('--train_portion', type=float, default=.9)
    parser.add_argument('--validate', action='store_true')
    parser.add_argument('--mjsynth', action='store_true')
    parser.add_argument('--imgh', type=int, default=100)
    parser.add_argument('--imgW', type=int, default=32)
    parser.add_argument('--workers', type=int, default=0,
                        help='image decoding processes feeding the generator (0 = the reference\'s single-threaded loader)')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.G >= 0:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(args.G))   # the reference falls back to CPU for G<0; this build has no CPU path
    import utils as U

    prng = RandomState(args.random_state)
    model = U.init_predictor(U.load_custom_model(args.model_path, model_name='/model.json', weights="/final_weights.h5"))
    classes = {ch: i for i, ch in enumerate(U.get_lexicon())}
    inverse_classes = {v: k for k, v in classes.items()}
    decoder = U.DecodeCTCPred(top_paths=1, beam_width=10, inverse_classes=inverse_classes)
    img_size = (args.imgh, args.imgW, 1)

    def walk():
        return np.array([os.path.join(dp, f) for dp, dn, fs in os.walk(args.image_path) for f in fs if re.search('png|jpeg|jpg', f)])

    if args.validate and args.mjsynth:
        fnames = np.array(U.parse_mjsynth(args.image_path, open(os.path.join(args.image_path, args.val_fname)).readlines()))
    else:
        fnames = walk()
        if args.validate:
            prng.shuffle(fnames)
            fnames = fnames[int(len(fnames) * args.train_portion):]
    if args.num_instances is not None:
        fnames = fnames[np.random.randint(0, len(fnames), min(args.num_instances, len(fnames)))]
    reader = U.Readf(img_size=img_size, normed=True, batch_size=args.batch_size, transform_p=0., classes=classes, max_len=args.max_len, workers=args.workers)
    length = len(fnames)
    bboxs = {}
    if args.boxes is not None:
        bboxs = pickle.load(open(args.boxes, "rb"))          # {image: [(word|None, x0, y0, x1, y1), ...]}
        half = len(bboxs) // 2
        bboxs = {os.path.join(args.image_path, k): v for i, (k, v) in enumerate(bboxs.items()) if i <= half}
        length = sum(len(v) for v in bboxs.values())
        fnames = list(bboxs.keys())
        if args.validate:
            y_true = np.array([reader.make_target(el[0]) for v in bboxs.values() for el in v], dtype=object)
    else:
        y_true = reader.get_labels(fnames)
    steps = -(-length // args.batch_size)
    print(" [INFO] Predicting... ")
    start = time.time()
    predicted = model.predict_generator(reader.run_generator(fnames, bboxs=bboxs, downsample_factor=2), steps=steps)
    print(" [INFO] %d images processed in %s sec. " % (len(fnames), round(time.time() - start, 2)))
    start = time.time()
    predicted_text = decoder.decode(predicted)[:length]
    print(" [INFO] %d predictions decoded in %s sec. " % (len(predicted), round(time.time() - start, 2)))
    if args.result_path is not None:
        import pandas as pd
        if len(fnames) != len(predicted_text):
            fnames = [f for f in bboxs for _ in range(len(bboxs[f]))]
        out_name = os.path.join(args.result_path, "prediction.csv")
        pd.DataFrame({"fname": fnames, "prediction": predicted_text}).to_csv(out_name)
        print(" [INFO] Prediction example: \n", predicted_text[:10])
        print(" [INFO] Result store in: ", out_name)
    if args.validate:
        print(" [INFO] Computing edit distance metric... ")
        start = time.time()
        true_text = [decoder.labels_to_text(y_true[i]) for i in range(len(y_true))]
        print(" [INFO] Example pairs (predicted, true): \n", list(zip(predicted_text[:10], true_text[:10])))
        ed = U.edit_distance(predicted_text, true_text)
        ned = U.normalized_edit_distance(predicted_text, true_text)
        print(" [INFO] edit distances calculated in %s sec. " % round(time.time() - start, 2))
        print(" [INFO] mean edit distance: %f ; normalized edit distance score: %f " % (ed, ned))


if __name__ == '__main__':
    main()
