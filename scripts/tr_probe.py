import ctypes, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "_trace", "libtr_probe.so"))
src = (np.arange(64)[:, None] * 200 + np.arange(160)[None, :]).astype(np.uint16)      # value = k*200 + col
s = torch.from_numpy(src.view(np.int16)).cuda(); d = torch.zeros(512, dtype=torch.int16, device="cuda")
rc = lib.tr_probe(ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
out = d.cpu().numpy().view(np.uint16).reshape(64, 8)
ok = True
for l in range(64):
    grp, li = l >> 4, l & 15
    k0, c = 8 * (grp >> 1), 16 * (grp & 1) + li
    exp = [(k0 + j) * 200 + c for j in range(8)]
    if list(out[l]) != exp:
        ok = False
        print("lane", l, "got", list(out[l]), "expected", exp)
        if l > 4: break
print("tr_probe semantics as assumed:", ok, "rc", rc)
