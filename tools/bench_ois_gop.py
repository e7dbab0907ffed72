import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
def timeit(fn, iters=4):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
W, H, pad = 1920, 1080 * 16, 64
plane = torch.randint(0, 256, (H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=dev)
pic = plane[pad:, pad:]
for bsize in (8, 16, 32, 64):
    blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
    xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
    modes, deltas = dsp.ois_candidates(bsize)
    ms = timeit(lambda: dsp.ois_search(pic, W + 2 * pad, W, H, xy, bsize, modes, deltas))
    px = len(blocks) * bsize * bsize * len(modes)
    print(json.dumps({"bsize": bsize, "blocks": len(blocks), "cand": len(modes), "ms": round(ms, 3), "ms_per_frame": round(ms / 16, 4), "Gpx_pred_per_s": round(px / ms / 1e6, 1)}), flush=True)
