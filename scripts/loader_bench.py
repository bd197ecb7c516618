#!/usr/bin/env python3
"""Host data path throughput: Readf.run_generator over JPEG text-line crops, single-threaded (the
reference's loop) and with worker processes.  CPU only.  usage: loader_bench.py [n_images] [workers ...]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np


def main():
    from PIL import Image
    from crnn_mi355x import data as D
    args = [int(a) for a in sys.argv[1:]]
    n = args[0] if args else 2048
    workers = args[1:] or [0, 4, 8, 16, 32, 64]
    tmp = tempfile.mkdtemp()
    rs = np.random.RandomState(0)
    words = ["hello", "world", "overfilled", "cellist", "amd", "mi355x", "ocr", "keras"]
    names = []
    for i in range(n):                                              # MJSynth-like crops: 31 px high, 60-124 px wide JPEGs
        a = (rs.rand(31, 60 + 8 * (i % 9), 3) * 255).astype(np.uint8)
        p = os.path.join(tmp, "%d_%s_%d.jpg" % (i, words[i % len(words)], i))
        Image.fromarray(a).save(p, quality=90)
        names.append(p)
    classes = {c: i for i, c in enumerate(D.get_lexicon())}
    out = {"images": n, "batch": 256, "host_cores": os.cpu_count(), "images_per_sec": {}}
    for w in workers:
        r = D.Readf(img_size=(100, 32, 1), max_len=23, normed=True, batch_size=256, classes=classes, workers=w)
        g = r.run_generator(names)
        next(g)
        t = time.time(); k = 0
        while time.time() - t < 3.0:
            next(g); k += 256
        out["images_per_sec"]["workers=%d" % w] = round(k / (time.time() - t), 1)
        print("workers=%d: %.0f img/s" % (w, out["images_per_sec"]["workers=%d" % w]), file=sys.stderr, flush=True)
        r.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
