"""Host data path: the batch contract that feeds the train step (reference utils.py:359-528).

Same names, arguments and quirks as the reference (`Readf`, `open_img`, `read_img`, `norm`, `parse_mjsynth`,
`get_lengths`, `get_lexicon`, `make_ohe`); OpenCV is replaced by PIL + NumPy (cv2 is not in the image):
  * cv2.resize(img, size, Image.LANCZOS) in the reference passes the PIL constant in the `dst` slot, so the
    effective interpolation is OpenCV's default INTER_LINEAR -> `resize_linear` below restates
    INTER_LINEAR's half-pixel-centre sampling (float arithmetic, round-half-up; OpenCV's 11-bit fixed point
    can differ by 1 grey level -- unpinned, cv2 cannot run here);
  * cv2.threshold(img, 127, 255, THRESH_BINARY) -> (img > 127) * 255; cv2.bitwise_not -> 255 - img.
The generator keeps the reference's behaviour of mutating and re-yielding the same arrays (utils.py:468-511).
"""
import os
import string

import numpy as np

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **k):
        return x


def get_lexicon(non_intersecting_chars=False):
    """utils.py:524-528"""
    if non_intersecting_chars:
        return list(set([i for i in '0123456789' + string.ascii_lowercase + 'AaBbDdEeFfGgHhLlMmNnQqRrTt' + '-']))
    return [i for i in '0123456789' + string.ascii_lowercase + '-']


def parse_mjsynth(path, names):
    """utils.py:412-413"""
    return [os.path.join(path, name.split()[0][2:]) for name in names]


def norm(image, mean, std):
    """utils.py:415-416"""
    return (image.astype('float32') - mean) / std


def get_lengths(names):
    """utils.py:518-522"""
    d = {}
    for name in tqdm(names, desc="getting words lengths"):
        d[name] = len(name.split("/")[-1].split("_")[1])
    return d


def make_ohe(y, nclasses):
    """utils.py:513-516"""
    ohe = np.zeros((len(y), nclasses))
    ohe[np.arange(len(y)), y.astype('int64')] = 1
    return ohe


def read_img(name):
    """utils.py:359-362: BGR decode -> gray (OpenCV weights .114 B + .587 G + .299 R, rounded)."""
    from PIL import Image
    img = np.array(Image.open(name).convert("RGB"), dtype=np.float64)
    gray = 0.299 * img[..., 0] + 0.587 * img[..., 1] + 0.114 * img[..., 2]
    return np.clip(np.floor(gray + 0.5), 0, 255).astype(np.uint8)


def resize_linear(img, dsize):
    """cv2.resize(img, (w, h)) with INTER_LINEAR semantics; img (H,W) uint8 -> (h,w) uint8."""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    src = img.astype(np.float64)

    def coords(n_out, n_in):
        s = (np.arange(n_out) + 0.5) * (n_in / float(n_out)) - 0.5
        i0 = np.floor(s).astype(np.int64)
        f = s - i0
        f = np.where(i0 < 0, 0.0, f); i0c = np.clip(i0, 0, n_in - 1)
        i1c = np.clip(i0 + 1, 0, n_in - 1)
        f = np.where(i0 >= n_in - 1, 0.0, f)
        return i0c, i1c, f

    y0, y1, fy = coords(h, H)
    x0, x1, fx = coords(w, W)
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    out = top * (1 - fy)[:, None] + bot * fy[:, None]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def open_img(img, img_size, p=.7):
    """utils.py:364-410 -- rotate (img[::-1].T), modal-value padding (random placement with prob. p),
    polarity inversion when the background is white, squash to (img_size[0], img_size[1])."""
    name = None
    if isinstance(img, str):
        name = img
        img = read_img(name)
    img = img[::-1].T
    val, counts = np.unique(img, return_counts=True)
    fill = val[np.where(counts == counts.max())[0][0]]
    if all([img.shape[0] <= img_size[0] // 2, img.shape[1] <= img_size[1] // 2]):
        img = resize_linear(img, (int(img.shape[1] * 1.5), int(img.shape[0] * 1.5)))
    if (img_size[1] - img.shape[1]) > 2:
        delta = img_size[1] - img.shape[1]
        r = round(np.random.uniform(0, 1), 1)
        if r < p and p > 0.:
            c = np.random.choice(list(range(2, delta)))
            start = np.full((img.shape[0], c - 1), fill)
            end = np.full((img.shape[0], delta - c), fill)
            img = np.concatenate([start, img, end], axis=1)
        else:
            img = np.concatenate([img, np.full((img.shape[0], delta), fill)], axis=1)
    if (img_size[0] - img.shape[0]) > 2:
        delta = img_size[0] - img.shape[0]
        r = round(np.random.uniform(0, 1), 1)
        if r <= p and p > 0.:
            c = np.random.choice(list(range(2, delta)))
            start = np.full((c - 1, img.shape[1]), fill)
            end = np.full((delta - c, img.shape[1]), fill)
            img = np.concatenate([start, img, end], axis=0)
        else:
            half = np.full(((delta) // 2, img.shape[1]), fill)
            img = np.concatenate([half, img, half], axis=0)
    img = img.astype(np.uint8)
    img_thrsh = np.where(img > 255 // 2, 255, 0).astype(np.uint8)
    val, counts = np.unique(img_thrsh, return_counts=True)
    if val[counts == counts.max()][0] == 255:
        img = (255 - img).astype(np.uint8)
    img = resize_linear(img, (img_size[1], img_size[0]))
    if name is not None:
        return img, name.split("/")[-1].split("_")[1].lower()
    return img, False


class Readf:
    """utils.py:418-511 -- same constructor, same generator contract:
    yields ({'the_input' (B,)+img_size float64, 'the_labels' (B,max_len) blank-padded, 'input_length' (B,1),
    'label_length' (B,1), 'source_str'}, {'ctc': zeros(B)})."""

    def __init__(self, img_size=(40, 40), max_len=30, normed=False, batch_size=32, classes={},
                 mean=118.24236953981779, std=36.72835353999682, transform_p=0.7):
        self.batch_size = batch_size
        self.transform_p = transform_p
        self.img_size = img_size
        self.normed = normed
        self.classes = classes
        self.max_len = max_len
        self.mean = mean
        self.std = std
        self.voc = list(self.classes.keys())
        if type(classes) == dict:
            self.blank = len(self.classes)

    def make_target(self, text):
        return np.array([self.classes[char] if char in self.voc else self.classes['-'] for char in text])

    def get_labels(self, names):
        Y_data = np.full([len(names), self.max_len], self.blank)
        for i, name in enumerate(names):
            img, word = open_img(name, self.img_size, p=self.transform_p)
            word = self.make_target(word)
            Y_data[i, 0:len(word)] = word
        return Y_data

    def get_blank_matrices(self):
        shape = (self.batch_size,) + tuple(self.img_size)
        X_data = np.empty(shape)
        Y_data = np.full([self.batch_size, self.max_len], self.blank)
        input_length = np.ones((self.batch_size, 1))
        label_length = np.zeros((self.batch_size, 1))
        return X_data, Y_data, input_length, label_length

    def run_generator(self, names, downsample_factor=2, bboxs={}):
        if bboxs:
            n_instances = sum([len(v) for v in bboxs.values()])
        else:
            bboxs = {name: [name] for name in names}
            n_instances = len(names)
        N = n_instances // self.batch_size
        rem = n_instances % self.batch_size
        i, n = 0, 0
        source_str = []
        X_data, Y_data, input_length, label_length = self.get_blank_matrices()
        while True:
            for name in names:
                if bboxs[name][0] == name:
                    _img, word = open_img(name, self.img_size, p=self.transform_p)
                else:
                    img = read_img(name)
                for bbox in bboxs[name]:
                    if bbox != name:
                        _img, __ = open_img(img[bbox[1]:bbox[3], bbox[2]:bbox[4]], self.img_size, p=self.transform_p)
                        word = bbox[0] if bbox[0] is not None else "-"
                    source_str.append(word)
                    word = self.make_target(word)
                    Y_data[i, 0:len(word)] = word
                    label_length[i] = len(word)
                    input_length[i] = (self.img_size[0] + 4) // downsample_factor - 2
                    if self.normed:
                        _img = norm(_img, self.mean, self.std)
                    X_data[i] = _img[:, :, np.newaxis]
                    i += 1
                    inputs = {'the_input': X_data, 'the_labels': Y_data, 'input_length': input_length,
                              'label_length': label_length, 'source_str': np.array(source_str)}
                    outputs = {'ctc': np.zeros([self.batch_size])}
                    if n == N and i == rem:
                        yield (inputs, outputs)
                    elif i == self.batch_size:
                        n += 1; i = 0
                        source_str = []
                        X_data, Y_data, input_length, label_length = self.get_blank_matrices()
                        yield (inputs, outputs)
