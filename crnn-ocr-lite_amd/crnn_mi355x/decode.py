"""DecodeCTCPred (utils.py:331-357): softmax maps -> text.  The reference loops over samples in Python and
builds one TF beam-search op + session.run per image (1.01 s/image, README.md:75); here the whole batch is
decoded by one launch of the HIP wavefront beam kernel (csrc/beam.hip), with TF-1.8 semantics including
merge_repeated=True ('cellist' decodes to 'celist', as in the reference's own screenshots)."""
import numpy as np


def labels_to_text(labels, inverse_classes=None):
    """utils.py:314-321: blank (== len(inverse_classes)) or -1 -> ''."""
    blank = len(inverse_classes)
    return "".join("" if (c == blank or c == -1) else str(inverse_classes[int(c)]) for c in labels)


class DecodeCTCPred:

    def __init__(self, top_paths=1, beam_width=5, inverse_classes=None, merge_repeated=True, greedy=False):
        self.top_paths = top_paths
        self.beam_width = beam_width
        self.inverse_classes = inverse_classes
        self.merge_repeated = merge_repeated     # TF-1.8 default, not overridden by Keras 2.2.2
        self.greedy = greedy                     # K.ctc_decode(greedy=True) variant (BASELINE config 2)

    def labels_to_text(self, labels):
        return labels_to_text(labels, self.inverse_classes)

    def decode_labels(self, result):
        """(N,T,C) softmax -> (N,T) int labels padded with -1 (device kernels; one launch per chunk)."""
        import torch
        from . import engine, native
        if self.beam_width < self.top_paths:
            self.beam_width = self.top_paths
        y = result if torch.is_tensor(result) else torch.from_numpy(np.ascontiguousarray(result, dtype=np.float32))
        out = []
        for lo in range(0, y.shape[0], 4096):
            chunk = y[lo:lo + 4096].cuda().contiguous()
            if self.greedy:
                B, T, C = chunk.shape
                lab = torch.empty((B, T), dtype=torch.int32, device=chunk.device)
                ln = torch.empty(B, dtype=torch.int32, device=chunk.device)
                native.check(native.lib().crnn_ctc_greedy_decode(engine._ptr(chunk), None, engine._ptr(lab), engine._ptr(ln), B, T, C,
                                                                 engine._stream()), "greedy")
            else:
                lab, ln, _ = engine.beam_decode(chunk, self.beam_width, self.merge_repeated)
            out.append(lab.cpu().numpy())
        return np.concatenate(out, 0) if out else np.zeros((0, 0), np.int32)

    def decode(self, result):
        return [self.labels_to_text(row) for row in self.decode_labels(result)]
