"""Host data path: the batch contract that feeds the train step (what reference utils.py:359-528 provides).

Same public names, signatures and observable behaviour as the reference (`Readf`, `open_img`, `read_img`,
`norm`, `parse_mjsynth`, `get_lengths`, `get_lexicon`, `make_ohe`), written against PIL + NumPy because
OpenCV is not in the image:
  * the reference's cv2.resize(img, size, Image.LANCZOS) passes the PIL constant in the `dst` positional slot,
    so the interpolation that actually runs is OpenCV's default INTER_LINEAR; `resize_linear`
    restates INTER_LINEAR's half-pixel-centre sampling in float arithmetic (OpenCV's 11-bit fixed point may
    differ by one grey level -- unpinned: cv2 cannot run here);
  * cv2.threshold(img, 127, 255, THRESH_BINARY) == (img > 127) * 255 and cv2.bitwise_not == 255 - img.
Quirks kept on purpose: the batch arrays are float64 and are re-yielded by reference; the
short last batch is emitted (full-size arrays, stale tail) only during the first pass because the batch
counter never resets; `input_length` is the constant (imgh+4)//downsample-2 for every sample.
"""
import os
import string

import numpy as np

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **k):
        return x

_ALNUM = '0123456789' + string.ascii_lowercase


def get_lexicon(non_intersecting_chars=False):
    """37 symbols 0-9 a-z '-' (utils.py:524-528)."""
    if non_intersecting_chars:
        return list(set(_ALNUM + 'AaBbDdEeFfGgHhLlMmNnQqRrTt' + '-'))
    return list(_ALNUM + '-')


def parse_mjsynth(path, names):
    """'./2425/1/115_Lube_45484.jpg 45484' -> path/2425/1/115_Lube_45484.jpg (utils.py:412-413)."""
    return [os.path.join(path, entry.split()[0][2:]) for entry in names]


def norm(image, mean, std):
    """(float32(image) - mean) / std  (utils.py:415-416)."""
    return (image.astype('float32') - mean) / std


def _word_of(fname):
    return fname.split("/")[-1].split("_")[1]


def get_lengths(names):
    """{file name: len(word encoded in the name)} (utils.py:518-522)."""
    return {name: len(_word_of(name)) for name in tqdm(names, desc="getting words lengths")}


def make_ohe(y, nclasses):
    """one-hot rows (utils.py:513-516)."""
    ohe = np.zeros((len(y), nclasses))
    ohe[np.arange(len(y)), y.astype('int64')] = 1
    return ohe


def read_img(name):
    """Decode to 8-bit gray with OpenCV's BGR2GRAY weights (utils.py:359-362)."""
    from PIL import Image
    rgb = np.array(Image.open(name).convert("RGB"), dtype=np.float64)
    gray = rgb @ np.array([0.299, 0.587, 0.114])
    return np.clip(np.floor(gray + 0.5), 0, 255).astype(np.uint8)


def _linear_taps(n_out, n_in):
    pos = (np.arange(n_out) + 0.5) * (n_in / float(n_out)) - 0.5
    lo = np.floor(pos).astype(np.int64)
    frac = pos - lo
    frac[lo < 0] = 0.0
    frac[lo >= n_in - 1] = 0.0
    return np.clip(lo, 0, n_in - 1), np.clip(lo + 1, 0, n_in - 1), frac


def resize_linear(img, dsize):
    """cv2.resize(img, (w, h)) with INTER_LINEAR semantics; (H,W) uint8 -> (h,w) uint8."""
    w, h = int(dsize[0]), int(dsize[1])
    src = img.astype(np.float64)
    y0, y1, fy = _linear_taps(h, src.shape[0])
    x0, x1, fx = _linear_taps(w, src.shape[1])
    rows0, rows1 = src[y0], src[y1]
    top = rows0[:, x0] * (1 - fx) + rows0[:, x1] * fx
    bot = rows1[:, x0] * (1 - fx) + rows1[:, x1] * fx
    out = top * (1 - fy)[:, None] + bot * fy[:, None]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def _modal_value(a):
    """First (= smallest) of the most frequent values, as utils.py:372-373 (np.unique + argmax)."""
    a = np.asarray(a)
    if a.dtype.kind in "ui" and a.size and a.min() >= 0 and a.max() < 65536:
        return a.dtype.type(np.bincount(a.ravel()).argmax())       # same answer, no sort
    vals, counts = np.unique(a, return_counts=True)
    return vals[np.argmax(counts)]


def _pad_axis(img, target, axis, fill, p, strict):
    """Pad `img` along `axis` up to `target` with the modal grey value (utils.py:379-400): with probability
    ~p the content is placed at a random offset, otherwise flush (axis 1) / centred (axis 0)."""
    delta = target - img.shape[axis]
    if delta <= 2:
        return img
    r = round(np.random.uniform(0, 1), 1)
    randomise = (r < p) if strict else (r <= p)

    def block(n):
        shape = (img.shape[0], n) if axis == 1 else (n, img.shape[1])
        return np.full(shape, fill)

    if randomise and p > 0.:
        c = np.random.choice(list(range(2, delta)))
        parts = [block(c - 1), img, block(delta - c)]
    elif axis == 1:
        parts = [img, block(delta)]
    else:
        parts = [block(delta // 2), img, block(delta // 2)]
    return np.concatenate(parts, axis=axis)


def open_img(img, img_size, p=.7):
    """Image (path or array) -> (img_size[0], img_size[1]) uint8 text-line with time on axis 0, plus the word
    encoded in the file name (or False).  Behaviour of utils.py:364-410: rotate via img[::-1].T, up-scale tiny
    crops by 1.5, modal-value padding, invert when the background is bright, squash (never crop) to size."""
    name = img if isinstance(img, str) else None
    if name is not None:
        img = read_img(name)
    img = img[::-1].T
    fill = _modal_value(img)
    if img.shape[0] <= img_size[0] // 2 and img.shape[1] <= img_size[1] // 2:
        img = resize_linear(img, (int(img.shape[1] * 1.5), int(img.shape[0] * 1.5)))
    img = _pad_axis(img, img_size[1], 1, fill, p, strict=True)
    img = _pad_axis(img, img_size[0], 0, fill, p, strict=False)
    img = img.astype(np.uint8)
    if _modal_value(np.where(img > 255 // 2, 255, 0)) == 255:
        img = (255 - img).astype(np.uint8)
    img = resize_linear(img, (img_size[1], img_size[0]))
    return img, (_word_of(name).lower() if name is not None else False)


def _chunk_seed(base, index):
    return (int(base) * 1000003 + int(index) * 7919 + 12345) & 0x7fffffff


def _decode_task(task):
    """Worker entry (module level: picklable under the spawn start method): one chunk of files, or one page with its
    boxes -> [(uint8 image, word)].  The padding offsets are drawn from a stream seeded per chunk, so what a chunk yields
    does not depend on which worker ran it."""
    kind, payload, img_size, p, seed = task
    np.random.seed(seed)
    if kind == "files":
        return [open_img(name, img_size, p=p) for name in payload]
    name, boxes = payload
    page = read_img(name)
    return [(open_img(page[b[1]:b[3], b[2]:b[4]], img_size, p=p)[0], (b[0] if b[0] is not None else "-")) for b in boxes]


class Readf:
    """Batch generator with the reference's constructor and contract (utils.py:418-511): yields
    ({'the_input' (B,)+img_size float64, 'the_labels' (B,max_len) filled with blank=len(classes),
      'input_length' (B,1), 'label_length' (B,1), 'source_str'}, {'ctc': zeros(B)}).

    Beyond the reference: `workers=N` (default 0 = the reference's single-threaded loop) decodes and pre-processes
    the images in N worker processes (spawn start method -- nothing of the GPU runtime is inherited), in order and with
    a bounded number of chunks in flight, so the generator keeps up with the device (one Python thread manages ~1 k
    images/s, the train step consumes 26 k/s per GPU).  Batches are identical to the serial loop when transform_p == 0;
    with random padding they are reproducible for a given `seed` whatever the worker count.  Scripts that use it
    need the usual `if __name__ == "__main__":` guard."""

    def __init__(self, img_size=(40, 40), max_len=30, normed=False, batch_size=32, classes={},
                 mean=118.24236953981779, std=36.72835353999682, transform_p=0.7, workers=0, seed=0, chunk=32):
        self.workers, self.seed, self.chunk = int(workers), int(seed), max(1, int(chunk))
        self._pool = None
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
        """characters -> class ids; unknown characters map to '-' (utils.py:435-436)."""
        dash = self.classes['-'] if '-' in self.classes else None
        return np.array([self.classes.get(ch, dash) for ch in text])

    def get_labels(self, names):
        Y = np.full([len(names), self.max_len], self.blank)
        for row, name in enumerate(names):
            _, word = open_img(name, self.img_size, p=self.transform_p)
            ids = self.make_target(word)
            Y[row, :len(ids)] = ids
        return Y

    def get_blank_matrices(self):
        X = np.empty((self.batch_size,) + tuple(self.img_size))
        Y = np.full([self.batch_size, self.max_len], self.blank)
        return X, Y, np.ones((self.batch_size, 1)), np.zeros((self.batch_size, 1))

    def _get_pool(self):
        if self._pool is None:
            import atexit, weakref
            import multiprocessing as mp
            self._pool = mp.get_context("spawn").Pool(self.workers)
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref().close())   # a pool left open is shut down gracefully at exit
        return self._pool

    def close(self):
        """Stop the worker processes (no-op for workers=0).  Graceful on purpose: the chunks already submitted finish
        (a bounded number) and the workers exit; Pool.terminate() can deadlock while results are still in flight."""
        pool, self._pool = self._pool, None
        if pool is not None:
            pool.close()
            pool.join()

    def _tasks(self, names, bboxs):
        """Endless stream of decode tasks in the reference's visiting order."""
        index = 0
        while True:
            run = []
            for name in names:
                boxes = bboxs[name]
                if boxes[0] == name:
                    run.append(name)
                    if len(run) == self.chunk:
                        yield ("files", run, self.img_size, self.transform_p, _chunk_seed(self.seed, index)); index += 1; run = []
                    continue
                if run:
                    yield ("files", run, self.img_size, self.transform_p, _chunk_seed(self.seed, index)); index += 1; run = []
                yield ("page", (name, boxes), self.img_size, self.transform_p, _chunk_seed(self.seed, index)); index += 1
            if run:
                yield ("files", run, self.img_size, self.transform_p, _chunk_seed(self.seed, index)); index += 1

    def _parallel_instances(self, names, bboxs):
        from collections import deque
        pool, tasks, pending = self._get_pool(), self._tasks(names, bboxs), deque()
        depth = 4 * self.workers
        while True:
            while len(pending) < depth:
                pending.append(pool.apply_async(_decode_task, (next(tasks),)))
            for item in pending.popleft().get():
                yield item

    def _instances(self, names, bboxs):
        """Endless stream of (uint8 image (H,W), word) in the reference's visiting order."""
        if self.workers > 0:
            yield from self._parallel_instances(names, bboxs)
            return
        while True:
            for name in names:
                boxes = bboxs[name]
                if boxes[0] == name:                      # whole-image sample, label from the file name
                    yield open_img(name, self.img_size, p=self.transform_p)
                    continue
                page = read_img(name)                     # (word|None, x0, y0, x1, y1) boxes on one page
                for box in boxes:
                    crop, _ = open_img(page[box[1]:box[3], box[2]:box[4]], self.img_size, p=self.transform_p)
                    yield crop, (box[0] if box[0] is not None else "-")

    def run_generator(self, names, downsample_factor=2, bboxs={}):
        if bboxs:
            total = sum(len(v) for v in bboxs.values())
        else:
            bboxs = {name: [name] for name in names}
            total = len(names)
        full_batches, remainder = divmod(total, self.batch_size)
        steps_in = (self.img_size[0] + 4) // downsample_factor - 2
        slot, emitted = 0, 0
        words = []
        X, Y, in_len, lab_len = self.get_blank_matrices()
        for pixels, word in self._instances(names, bboxs):
            words.append(word)
            ids = self.make_target(word)
            Y[slot, :len(ids)] = ids
            lab_len[slot] = len(ids)
            in_len[slot] = steps_in
            X[slot] = (norm(pixels, self.mean, self.std) if self.normed else pixels)[:, :, np.newaxis]
            slot += 1
            tail = emitted == full_batches and slot == remainder
            if not tail and slot != self.batch_size:
                continue
            batch = ({'the_input': X, 'the_labels': Y, 'input_length': in_len, 'label_length': lab_len,
                      'source_str': np.array(words)}, {'ctc': np.zeros([self.batch_size])})
            if tail:
                yield batch            # short tail of the first pass: same arrays, stale rows beyond `slot`
            else:
                emitted += 1
                slot, words = 0, []
                X, Y, in_len, lab_len = self.get_blank_matrices()
                yield batch
