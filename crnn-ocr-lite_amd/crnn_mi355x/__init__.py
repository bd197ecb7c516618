"""crnn_mi355x -- MI355X-native (gfx950) CRNN-OCR hot path behind the reference's Python surface.

Host code is plumbing (torch for device memory/streams/torch.distributed); all arithmetic on the path
runs in libcrnn_mi355x.so (hand-written HIP).  There is no CPU fallback."""
