import os, sys, json
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
n = 1 << 20
src = torch.randint(0, 256, (n, 16, 16), dtype=torch.uint8, device=dev)
for (rh, rw, sw, sh, tag) in ((23, 23, 8, 8, "C3 23x23 pitch23"), (23, 32, 8, 8, "C3 pitch32 aligned"), (23, 23, 1, 1, "1 cand pitch23"), (23, 32, 4, 1, "4 cand pitch32"), (23,32,8,2,"16 cand"), (31, 32, 16, 16, "256 cand 16x16 search")):
    ref = torch.randint(0, 256, (n, rh, rw), dtype=torch.uint8, device=dev)
    for q in (0, 1):
        dsp.lib.svt_hip_tune(b"no_qsad", q)
        ms = timeit(lambda: dsp.sad_search(src, ref, sw, sh))
        print(tag, "old" if q else "qsad", round(ms, 4), "ms", flush=True)
    del ref
