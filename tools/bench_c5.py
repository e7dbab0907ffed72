#!/usr/bin/env python3
"""BASELINE.json configs[4]: 3840x2160 yuv420p10 (uint16 samples, 10-bit), 240 frames in 8 GOPs of 30; GOP g is owned by
rank g mod G (one process per GPU, torch.distributed; nccl == RCCL).  Per frame, ONE svt_hip_encode_recon_frame call runs the
encode-pass chain (residual -> FwdTxfm2d -> quantise / dequantise -> InvTxfm2d -> reconstruction, bd 10) for every CU size
of SURVEY 8(d): luma 64/32/16/8/4 and both chroma planes at half the side.  No data-path collective: a rank holds only its
own GOP's planes (one GOP resident at a time: 30 x 2 x 24.9 MB); the only exchange is the all-reduce of a 4-word digest
(cidana-svt-av1_amd/sharding.py) at the end, and the barrier / max-reduce around the timed region.

    python tools/bench_c5.py                      # 1 GPU: all 8 GOPs, one after the other
    python tools/bench_c5.py --gpus N             # starts its own N ranks (one per GPU) as child processes
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_c5.py --gpus N   # or under a launcher
    options: --frames 240 --gop 30 --width 3840 --height 2160 --sizes 64,32,16,8,4 --qindex 120
             --rehearse  (all ranks on cuda:0 with the gloo backend: exercises the N > 1 code path on a one-GPU box)

Prints one JSON line on rank 0: frames/s, pixels/s, GB/s at 7 B/px x 2 (16-bit samples: src + pred in, recon out = 6 B/px,
qcoeff 4 B/px, eob) of the kept outputs, the all-reduced digest and the per-rank GOP map."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--gop", type=int, default=30)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--sizes", default="64,32,16,8,4")
    ap.add_argument("--qindex", type=int, default=120)
    ap.add_argument("--stack", type=int, default=1, help="pictures of a GOP handed to the device in ONE call (<= 65535 / height)")
    ap.add_argument("--single-launch", type=int, default=-1, help="1 / 0: force svt_hip_tune(frame_single_launch); -1: library default")
    ap.add_argument("--rehearse", action="store_true")
    ap.add_argument("--spread-all", dest="spread_outputs", action="store_const", const="all", help="source and prediction planes apart as well")
    ap.add_argument("--interleaved-outputs", dest="spread_outputs", action="store_false",
                    help="allocate every group's qcoeff and recon arrays one after the other (rounds 1 - 3); default: all qcoeff arrays, a temporary "
                         "32 GiB spacer, all recon arrays - the placement rule of DESIGN 3, + 4.5 % on this config")
    ap.add_argument("--json-out", default=None)
    args = ap.parse_args()

    # bare `python tools/bench_c5.py --gpus N`: this process only starts the N ranks (before torch / HIP are loaded)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from cidana_svt_av1_amd import launcher
    if launcher.needs_spawn(args.gpus):
        sys.exit(launcher.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = launcher.check_world(args.gpus)     # exits non-zero when WORLD_SIZE and --gpus disagree

    import torch
    from cidana_svt_av1_amd import frames, sharding
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dsp = pkg.SvtHipDsp(local_rank)
    if args.single_launch >= 0:
        assert dsp.lib.svt_hip_tune(b"frame_single_launch", args.single_launch) == 0
    qt = pkg.tables.quant_tables(10)
    qrow = {k: v[args.qindex].copy() for k, v in qt.items()}
    sizes = tuple(int(s) for s in args.sizes.split(","))
    W, H = args.width, args.height
    n_gops = (args.frames + args.gop - 1) // args.gop
    mine = sharding.gops_of_rank(n_gops, rank, world)

    def make_gop(g):
        """synthetic planes of one GOP, generated on the device (seed = 13596 + global frame index)"""
        passes, pics = [], []
        for f in range(g * args.gop, min((g + 1) * args.gop, args.frames)):
            gen = torch.Generator(device=dev); gen.manual_seed(13596 + f)
            src, pred = {}, {}
            for name, (ph, pw) in (("Y", (H, W)), ("U", (H // 2, W // 2)), ("V", (H // 2, W // 2))):
                s = torch.randint(0, 1024, (ph, pw), dtype=torch.int16, device=dev, generator=gen)
                p = (s + torch.randint(-64, 65, (ph, pw), dtype=torch.int16, device=dev, generator=gen)).clamp_(0, 1023)
                src[name], pred[name] = s, p
            pics.append((src, pred))
        k = max(1, min(args.stack, 65535 // H))
        for i in range(0, len(pics), k):
            grp = pics[i:i + k]
            if len(grp) == 1:
                passes.append(frames.FramePass(dsp, pkg, grp[0][0], grp[0][1], luma_sizes=sizes, is_16bit=True, spread_outputs=args.spread_outputs))
            else:                                   # the pictures of this path are independent: one call for the stack
                src = {n: torch.stack([q[0][n] for q in grp]) for n in ("Y", "U", "V")}
                pred = {n: torch.stack([q[1][n] for q in grp]) for n in ("Y", "U", "V")}
                passes.append(frames.FramePass(dsp, pkg, src, pred, luma_sizes=sizes, is_16bit=True, spread_outputs=args.spread_outputs))
                passes[-1].nframes = len(grp)
        return passes

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    digest = torch.zeros(4, dtype=torch.int64, device=dev)
    # warm-up: one frame of the first owned GOP (kernel code objects, internal streams)
    if mine:
        warm = make_gop(mine[0])[:1]
        warm[0].run(qrow)
        torch.cuda.synchronize()
        del warm
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    busy = 0.0
    frames_done = 0
    px_done = 0
    for g in mine:
        passes = make_gop(g)                       # generation is outside the kernel-time sum but inside the wall clock
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for p in passes:
            p.run(qrow)
        e1.record()
        torch.cuda.synchronize()
        busy += e0.elapsed_time(e1) * 1e-3
        for p in passes:
            digest += p.digest()
            px_done += p.pixels
        frames_done += sum(getattr(p, "nframes", 1) for p in passes)
        del passes
    digest[2] %= sharding.DIGEST_MOD
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tot = sharding.allreduce_digest(digest.cpu().numpy(), None if args.rehearse else (dev if world > 1 else None), checksum_lanes=(2,))
    stats = torch.tensor([busy, wall, float(frames_done), float(px_done)], dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        mx = stats.clone() if args.rehearse else stats.to(dev)
        sm = mx.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        busy_max, wall_max = float(mx[0]), float(mx[1])
        frames_all, px_all = float(sm[2]), float(sm[3])
    else:
        busy_max, wall_max, frames_all, px_all = busy, wall, float(frames_done), float(px_done)
    if rank == 0:
        bytes_per_px = 2 + 2 + 2 + 4            # u16 src + pred in, u16 recon + i32 qcoeff out (eob is noise)
        out = {"config": "configs[4]: %dx%d yuv420p10, %d frames, GOP %d, sizes %s, encode-pass chain bd 10, %d picture(s) per call" % (W, H, args.frames, args.gop, args.sizes, max(1, min(args.stack, 65535 // H))),
               "n_gpus": world, "rehearsal_all_ranks_on_one_gpu": bool(args.rehearse), "gops": n_gops,
               "gop_owner": {str(g): sharding.gop_owner(g, world) for g in range(n_gops)},
               "frames": frames_all, "pixel_passes": px_all, "kernel_seconds_max_rank": busy_max, "wall_seconds_max_rank": wall_max,
               "frames_per_s_kernel": frames_all / busy_max if busy_max else None, "Gpx_per_s_kernel": px_all / busy_max / 1e9 if busy_max else None,
               "GBps_kernel_at_%dB_per_px" % bytes_per_px: px_all * bytes_per_px / busy_max / 1e9 if busy_max else None,
               "frac_of_8TBps_x_ngpus": px_all * bytes_per_px / busy_max / 8e12 / world if busy_max else None,
               "digest_blocks_eob_qchk_recon": [int(v) for v in tot], "device": dsp.device_name()}
        line = json.dumps(out)
        print(line, flush=True)
        if args.json_out:
            open(args.json_out, "w").write(line + "\n")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
