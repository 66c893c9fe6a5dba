#!/usr/bin/env python3
"""bench.py -- headline benchmark of the RayTracinGO hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W          (default: N=1, K=20, W=5)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): cornell 1920x1080 --mode=path --sample=4 (16 spp), progressive frames.
A step = one frame = one megakernel launch per GPU over that GPU's row bands (+ for N > 1 the RCCL gather of the 8-bit
bands to rank 0, the only exchange on the path).  Inputs (scene, LBVH, accumulation buffer) are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
FP32_PEAK_TFLOPS = 157.3  # vector fp32 (non-MFMA) peak, same guide
PROFILE_STALE_REL = 0.05  # a committed PMC profile speaks for this run only while its kernel time is within 5 % of the one measured here


def committed_profile(scene, W, H, mode, N):
    """The latest committed PMC profile of this workload (profiles/*/summary.json, written by tools/profile_round.sh from separate
    rocprofv3 --pmc passes): HBM bytes per launch, and what binds the kernel -- vector issue slots filled, lanes live in them,
    vector / scalar instructions per traversed ray.  None when there is none."""
    import glob
    best = None
    key = "%s %dx%d --mode=%s --sample=%d" % (scene, W, H, mode, N)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "summary.json"))):   # (by name: rNNx tags sort by round)
        try:
            j = json.load(open(f))
            if key in j["bench"]["config"]["workload"] and "hbm_traffic_bytes_per_launch" in j:
                best = (j, os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def usable_cores():
    """threads the CPU baseline may really use: the cgroup CPU quota when there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(scene, width, height, sample, path, seconds_cap=30.0):
    """The oracle (scalar C restatement, canonical LBVH, OpenMP over rows) timed on this host's cores: a reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build()
    sc = O.scene(scene, width, height)
    cores = usable_cores()
    # bounded sample: time a 1/16-area crop first, then the largest centred window expected to fit the cap
    wq, hq = width // 4, height // 4
    win = ((width - wq) // 2, (height - hq) // 2, wq, hq)
    t0 = time.perf_counter()
    acc, _, c = O.render(sc, O.frame(width, height, sample, 0, path=path, window=win, mode=1, threads=cores))
    dt = time.perf_counter() - t0
    sample_desc = "centre crop %dx%d of frame 0" % (wq, hq)
    rate = c["rays_total"] / dt
    # centre crops are denser than the full frame; if the full frame fits the cap, use it instead
    full_rays_est = width * height * sample * sample * 2.3
    if full_rays_est / rate < seconds_cap * 0.6:
        t0 = time.perf_counter()
        acc, _, c = O.render(sc, O.frame(width, height, sample, 0, path=path, mode=1, threads=cores))
        dt = time.perf_counter() - t0
        sample_desc = "full frame 0 (%dx%d, %d spp)" % (width, height, sample * sample)
        win = (0, 0, width, height)
    return {"value": round(c["rays_total"] / dt / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": sample_desc, "seconds": round(dt, 2), "rays": c["rays_total"],
            "V": round(c["node_visits"] / c["rays_total"], 3), "T": round(c["prim_tests"] / c["rays_total"], 3),
            "h": round(c["hits"] / c["rays_total"], 4)}, acc, win


def parity_against(ref_accum, win, gpu_accum):
    """the checker's verdict on the GPU's frame 0 over the window the CPU baseline rendered (tolerance of tests/parity.py)"""
    import numpy as np
    x0, y0, w, h = win
    g = gpu_accum[y0:y0 + h, x0:x0 + w, :3].astype(np.float64)
    r = np.asarray(ref_accum)[..., :3].astype(np.float64)
    ok = (np.abs(g - r) <= 1e-4 * np.maximum(1.0, np.abs(r))).all(axis=-1)
    return {"window": list(win), "pixels": int(ok.size), "frac_within_1e-4": round(float(ok.mean()), 6),
            "frac_bit_exact": round(float((gpu_accum[y0:y0 + h, x0:x0 + w, :3] == np.asarray(ref_accum)[..., :3]).all(axis=-1).mean()), 6),
            "mean_rel_diff": float(abs(g.mean() - r.mean()) / max(r.mean(), 1e-30))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sample", type=int, default=4)
    ap.add_argument("--mode", default="path", choices=["path", "distributed"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spin-up", type=int, default=100, help="untimed launches into scratch buffers before the warm-up steps (GPU clocks)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the per-frame band gather (diagnostic)")
    ap.add_argument("--launches-per-frame", type=int, default=0, choices=[0, 1, 2],
                    help="a rank renders its share as 1 launch, or as 2 half-share launches on two contexts / HIP streams so that "
                         "the tail and launch latency of one overlap the body of the other (0 = 1: it stopped paying with round 3's kernel)")
    ap.add_argument("--single-rank-collectives", action="store_true",
                    help="developer check on a 1-GPU box: run the N>1 machinery (RCCL process group, comm stream, triple-buffered "
                         "bands, gather + de-interleave every frame) with a world of one rank. The JSON line is marked.")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="developer rehearsal of the N>1 code path on a 1-GPU box: every rank uses cuda:0 and the gather goes "
                         "through gloo on host copies. The JSON line is marked and is NOT a measurement.")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print there (RCCL's version banner at start-up) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracingo_amd import bands, capi, scene as hscene

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the hot path has no CPU fallback")
    rehearse = args.rehearse_on_one_gpu
    multi = world > 1 or args.single_rank_collectives   # the N>1 machinery is on
    if args.single_rank_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            try:   # RCCL's internal stream at high priority: its small kernels should not queue behind the megakernel
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                dist.init_process_group("nccl", device_id=dev, pg_options=opts)
            except Exception:
                dist.init_process_group("nccl", device_id=dev)

    W, H, N = args.width, args.height, args.sample
    path = args.mode == "path"
    band_h = 4
    t = hscene.tables(args.scene, W, H)          # C++ host (engine::host::Scene) -> flattened tables
    # S launches per frame and rank: the rank's share is cut into S sub-shares (virtual ranks v = rank*S + i of V = world*S in the
    # band interleave), each with its own context and HIP stream.  A short launch ends with a tail in which few waves are still
    # busy, and starts with ~13 us of launch latency; with two launches in flight one's tail and start overlap the other's body
    # Round 1-2 kernels: a 1/8 share of the 1080p frame took 0.220 ms as one launch and 0.199 ms as two, hence S = 2 from N = 4 on.
    # With round 3's kernel it no longer pays (profiles/r03f: 1/8 share 0.169 ms as one launch, 0.175 as two; 1/4: 0.275 / 0.295),
    # and two launches, their events and stream waits cost this driver 0.19 ms of host time per step against 0.08 for one -- more
    # than the share's kernel.  One launch per frame and rank, unless asked otherwise.
    S = args.launches_per_frame if args.launches_per_frame else 1
    V = world * S
    ctxs = []
    for i in range(S):
        c = capi.Context(local_rank)
        c.set_scene(t["type"], t["M"], t["mat"], t["aabb"])
        c.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
        c.set_background(t["bg"])
        c.set_lights(t["lights"])
        ctxs.append(c)
    ctx = ctxs[0]

    rows_pad = bands.max_local_rows(H, band_h, V)      # rows of one sub-share, padded to the largest
    sub_px = rows_pad * W
    accum = torch.zeros((S * rows_pad, W, 4), dtype=torch.float32, device=dev)
    image = torch.zeros((S * rows_pad, W, 4), dtype=torch.uint8, device=dev)
    scratch_a = torch.zeros((max(S * rows_pad, H), W, 4), dtype=torch.float32, device=dev)
    scratch_i = torch.zeros((max(S * rows_pad, H), W, 4), dtype=torch.uint8, device=dev)
    # explicit (non-default) HIP streams carry the megakernels AND everything torch enqueues (the RCCL gather waits on
    # them), so kernel -> gather ordering is by stream; the ABI treats a NULL stream as "use the context's own"
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    stream = streams[0]
    torch.cuda.set_stream(stream)
    for c, st_ in zip(ctxs, streams):
        assert st_.cuda_stream != 0
        c.set_stream(st_.cuda_stream)
    full_image = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    row_index = bands.full_row_index(H, band_h, V, rows_pad, dev) if (rank == 0 and multi) else None
    row_index_host = bands.full_row_index(H, band_h, V, rows_pad) if (rank == 0 and rehearse) else None

    def frame(f, i=0, stats=False):
        # N > 1: leave 8 CUs' worth of workgroup slots free so that RCCL's kernels can run beside the persistent megakernel
        return capi.make_frame(W, H, N, f, path, False, None, (band_h, V, rank * S + i), stats=stats,
                               reserve_cus=(8 if multi and not stats else 0))

    def bind(i, acc_t, img_t):
        ctxs[i].bind_output(acc_t.data_ptr() + i * sub_px * 16, img_t.data_ptr() + i * sub_px * 4, sub_px)

    # N > 1: the 8-bit bands are triple-buffered so that the gather of frame f (comm stream -> RCCL) overlaps the
    # megakernels of frame f+1 (compute streams).  Every frame is still gathered and de-interleaved on rank 0 inside the
    # timed region; the accumulation buffer is single (frame f+1 reads what frame f wrote, same stream).
    NBUF = 3
    images = [image] + [torch.zeros_like(image) for _ in range(NBUF - 1)] if multi else [image]
    comm = torch.cuda.Stream(dev, priority=-1) if multi else None
    gathered = [None] * NBUF   # event: the gather that last read images[b] has completed
    gather_ws = (torch.empty((world * S * rows_pad, W, 4), dtype=torch.uint8, device=dev) if (multi and rank == 0) else None)

    def step(f):
        b = f % NBUF if multi else 0
        for i in range(S):
            if multi and gathered[b] is not None:
                streams[i].wait_event(gathered[b])        # WAR: do not overwrite a band buffer a gather is still reading
            bind(i, accum, images[b])
            ctxs[i].launch(frame(f, i))
        if multi and not args.no_gather:
            for i in range(S):
                done = torch.cuda.Event()
                done.record(streams[i])
                comm.wait_event(done)
            with torch.cuda.stream(comm):
                if rehearse:
                    comm.synchronize()
                    full = bands.gather_bands(images[b].cpu(), H, band_h, dist, dst=0, row_index=row_index_host)
                    if rank == 0:
                        full_image.copy_(full)
                else:
                    bands.gather_bands(images[b], H, band_h, dist, dst=0, out=full_image, row_index=row_index, workspace=gather_ws)
                ev = torch.cuda.Event()
                ev.record(comm)
            gathered[b] = ev

    def sync_all():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def stats_sum():
        tot = None
        for c in ctxs:
            st_ = c.stats()
            if tot is None:
                tot = dict(st_)
            else:
                for k in ("rays_total", "rays_occlusion", "node_visits", "prim_tests", "hits", "rays_culled", "total_launch_ms", "launches"):
                    tot[k] += st_[k]
        return tot

    # ---- V, T, h of the workload's TRAVERSED rays (instrumented kernel with the same background culling as the timed one,
    #      collect_stats = 2; untimed, into scratch buffers so the accumulation is untouched)
    for i in range(S):
        bind(i, scratch_a, scratch_i)
        ctxs[i].reset_stats()
        ctxs[i].launch(frame(args.warmup, i, stats=2))
    for c in ctxs:
        c.sync()
    st = stats_sum()
    rdev = torch.device("cpu") if rehearse else dev   # gloo reduces host tensors
    cnt = torch.tensor([st["rays_total"] - st["rays_culled"], st["node_visits"], st["prim_tests"], st["hits"], st["rays_occlusion"]], dtype=torch.float64, device=rdev)
    if multi:
        dist.all_reduce(cnt)
    rays_s, nodes_s, tests_s, hits_s, occl_s = [float(x) for x in cnt.tolist()]   # rays_s: traversed rays of one frame
    Vbar, Tbar, hbar = nodes_s / rays_s, tests_s / rays_s, hits_s / rays_s
    A_ray = 32 + 32 + 32 * Vbar + 64 * Tbar + 40 * hbar
    A_px = 36                                             # frame > 0: float4 read + float4 write + uchar4 write

    # ---- clocks: a GPU that has been idle runs its first milliseconds below its sustained clock, and W = 5 warm-up frames are 5 ms.
    #      Untimed launches into the scratch buffers first (the accumulation buffer and the frame numbers of the run are untouched),
    #      so that a short run (--steps 20) measures the same kernel time as a long one and as rocprofv3 does.
    for i in range(S):
        bind(i, scratch_a, scratch_i)
    for k in range(args.spin_up):
        for i in range(S):
            ctxs[i].launch(frame(k, i))
    for c in ctxs:
        c.sync()

    # ---- warmup + timed region
    for f in range(args.warmup):
        step(f)
    sync_all()
    for c in ctxs:
        c.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync_all()
    dt = time.perf_counter() - t0
    st = stats_sum()
    tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
    rr = torch.tensor([float(st["rays_total"]), float(st["rays_culled"])], dtype=torch.float64, device=rdev)
    # S == 1: the megakernel's average duration by HIP events.  S == 2: the two launches of a frame overlap, so their event
    # durations do too; the rank's kernel time per frame is then its wall time per frame
    km = torch.tensor([float(st["total_launch_ms"]) / max(st["launches"], 1) if S == 1 else dt / args.steps * 1e3], dtype=torch.float64, device=rdev)
    if multi:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    rays_total, rays_culled = [float(x) for x in rr.tolist()]
    kernel_ms = float(km.item())          # slowest rank's average megakernel duration (HIP events on the launch stream)

    if rank == 0:
        prof = committed_profile(args.scene, W, H, args.mode, N) if world == 1 else None
        ms_per_step = dt / args.steps * 1e3
        rays_per_launch = rays_total / args.steps                      # optixTrace equivalents per frame, all ranks
        trav_per_launch = (rays_total - rays_culled) / args.steps      # ... of which a traversal answered
        kernel_s = kernel_ms * 1e-3
        # HBM roofline: with the scene resident in LDS, the bytes the algorithm moves through HBM are the framebuffer's
        # (A_px = 36 B per pixel-frame on frames > 0: float4 accumulation read + write, uchar4 image write)
        hbm_alg = W * H * A_px / world
        achieved = hbm_alg / kernel_s / 1e9 if kernel_s > 0 else 0.0
        # SURVEY 8d's per-ray byte model, on the traversed rays: node and primitive records and materials -- bytes this design
        # serves from LDS and the scalar cache, never from HBM, so they are priced against the LDS peak, not against HBM's
        ray_bytes = trav_per_launch * A_ray / world
        binding = None
        traffic = None
        if prof:
            # Everything in `binding` and `traffic` comes from the committed rocprofv3 --pmc profile of this workload, NOT from this run
            # (counters need the profiler); it is printed only while that profile's kernel time agrees with the one measured here.
            j = prof[0]
            c = j.get("pmc_per_launch", {})
            prof_ms = j.get("kernel_trace", {}).get("avg_ms", 0.0)
            stale = not (prof_ms > 0 and abs(kernel_ms - prof_ms) <= PROFILE_STALE_REL * prof_ms)
            if stale:
                binding = {"withheld": "the committed profile (%s, kernel %.4f ms at %s) is more than %d %% away from this run's kernel time (%.4f ms): re-profile with tools/profile_round.sh"
                                       % (prof[1], prof_ms, j.get("head", "unknown commit"), int(PROFILE_STALE_REL * 100), kernel_ms)}
            else:
                traffic = round(j["hbm_traffic_bytes_per_launch"])
                prof_trav = j["bench"]["config"].get("rays_per_frame", 0) - j["bench"]["config"].get("rays_culled_per_frame", 0)
                lanes = j.get("valu_lane_utilisation", 0.0)
                flops = (c.get("SQ_INSTS_VALU_ADD_F32", 0.0) + c.get("SQ_INSTS_VALU_MUL_F32", 0.0) + 2.0 * c.get("SQ_INSTS_VALU_FMA_F32", 0.0)) * 64.0 * lanes
                binding = {"bound": "valu_issue",
                           "valu_issue_busy": round(j.get("valu_issue_busy", 0.0), 4),
                           "valu_lane_utilisation": round(lanes, 4),
                           # share of the chip's vector lane-slots that do work: issue slots filled x lanes live in them
                           "frac_binding": round(j.get("valu_issue_busy", 0.0) * lanes, 4),
                           "valu_insts_per_traversed_ray": round(c.get("SQ_INSTS_VALU", 0.0) / prof_trav * 64.0, 1) if prof_trav else None,
                           "salu_insts_per_traversed_ray": round(c.get("SQ_INSTS_SALU", 0.0) / prof_trav * 64.0, 1) if prof_trav else None,
                           "insts_unit": "wave instructions per 64 traversed rays (one ray per lane)",
                           # counted, not modelled: (ADD + MUL + 2 FMA f32 wave instructions) x 64 lanes x the share of lanes live / kernel time
                           "fp32_counted_tflops": round(flops / (prof_ms * 1e-3) / 1e12, 2) if prof_ms > 0 else None,
                           "fp32_peak_tflops": FP32_PEAK_TFLOPS,
                           "profile_kernel_ms": round(prof_ms, 4),
                           "profile_head": j.get("head", "unknown"),
                           "source": prof[1],
                           "basis": "committed rocprofv3 --pmc profile of the same workload (separate passes); this run measured kernel_ms, ms_per_step and the ray counts"}
        out = {
            "metric": "Mray/s + ms/frame, %s %dx%d %s spp=%d" % (args.scene, W, H, args.mode, N * N),
            # rays a traversal answered per second; the primary rays of pixels outside the scene's screen rectangle (the reference
            # traces them, they miss) are answered without a walk and are NOT counted here: config.mrays_incl_culled_primary has them
            "value": round((rays_total - rays_culled) / dt / 1e6, 2), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (procedural scene of the reference, fixed RNG seeds)",
            "config": {"workload": "%s %dx%d --mode=%s --sample=%d (%d spp), progressive frames %d..%d" %
                                   (args.scene, W, H, args.mode, N, N * N, args.warmup, args.warmup + args.steps - 1),
                       "primitives": int(len(t["type"])), "tiling": "4-row bands interleaved over %d GPU(s)" % world,
                       "launches_per_frame_per_gpu": S,
                       "gather": "RCCL gather of uchar4 bands to rank 0 every frame, overlapped with the next frame's kernel" if (multi and not args.no_gather) else "none",
                       "rays_per_frame": int(round(rays_per_launch)),
                       "rays_culled_per_frame": int(round(rays_culled / args.steps)),
                       "rays_traversed_per_frame": int(round(trav_per_launch)),
                       "mrays_incl_culled_primary": round(rays_total / dt / 1e6, 2),
                       # what the timed launches ran (rank 0; rtgo_stats.last_variant): the launch-time trial of the spin-up chose it
                       "kernel_variant": {"walk": ("canonical LBVH (beyond the far-field guard)" if st["last_variant"] & 4 else
                                                   "uniform grid" if st["last_variant"] & 16 else
                                                   "tree, big-primitive threshold 15 %" if st["last_variant"] & 2 else "tree, big-primitive threshold 36 %"),
                                          "loop": "streaming" if st["last_variant"] & 1 else "lock-step",
                                          "trial_launches_in_timed_region": int(st["launches_trial"])}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch, PMC (2*FETCH_SIZE+WRITE_SIZE)*1024",
                         "traffic_source": (prof[1] if traffic is not None else None),
                         "algorithmic_bytes_per_launch": int(hbm_alg),
                         "algorithmic_basis": "SURVEY 8d A_px = %d B per pixel-frame x %d x %d pixels: the scene (<= 512 primitives) is resident in LDS, "
                                              "so the framebuffer is all the algorithm moves through HBM" % (A_px, W, H),
                         "kernel": "rtgo::render_kernel<%s,false>" % ("true" if path else "false"),
                         "kernel_ms": round(kernel_ms, 4),
                         "kernel_ms_basis": ("HIP events around each launch, averaged (slowest rank)" if S == 1 else
                                             "wall time per frame of the slowest rank: its two launches per frame overlap, and so do their event durations"),
                         # what binds the kernel: vector issue (PMC of the committed profile of this workload)
                         "binding": binding,
                         "on_chip_ray_model": {"A_ray_bytes": round(A_ray, 1), "V": round(Vbar, 3), "T": round(Tbar, 3), "h": round(hbar, 4),
                                               "counted_over": "traversed rays (instrumented canonical-LBVH launch with the timed kernel's background culling)",
                                               "bytes_per_launch": int(ray_bytes),
                                               "rate_GBps": round(ray_bytes / kernel_s / 1e9, 1) if kernel_s > 0 else 0.0,
                                               # SURVEY 8d's floor: 32 + 32 + 64 + 40 = 168 B per ray (one primitive test, no nodes)
                                               "A_ray_min_bytes": 168, "rate_min_GBps": round(trav_per_launch * 168.0 / world / kernel_s / 1e9, 1) if kernel_s > 0 else 0.0,
                                               "served_from": "LDS and the scalar cache (not HBM): SURVEY 8d's byte model priced on the canonical LBVH's counts, which the timed kernel does not walk -- a workload description, not an achieved rate of this kernel"},
                         "note": "frac is small by design: the kernel is bound by vector issue (binding), not by HBM"},
        }
        if args.single_rank_collectives:
            out["single_rank_collectives"] = "N>1 machinery with a world of one rank (developer check): NOT the N=1 measurement"
        if rehearse:
            out["rehearsal"] = "all ranks on cuda:0, gloo gather through host memory: NOT a measurement"
        if rehearse or args.single_rank_collectives:
            # correctness of the N>1 path: the gathered 8-bit frame must equal a whole-image render of the same frames
            whole_a = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
            whole_i = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
            ctx.bind_output(whole_a.data_ptr(), whole_i.data_ptr(), H * W)
            for f in range(args.warmup + args.steps):
                ctx.launch(capi.make_frame(W, H, N, f, path, False, None, (band_h, 1, 0)))
            ctx.sync()
            out["gathered_frame_matches_single_launch"] = bool(torch.equal(whole_i, full_image))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_acc, ref_win = cpu_baseline(args.scene, W, H, N, path)
            # the same frame 0 on the GPU (untimed, scratch buffers), held against what the oracle just rendered
            ctx.bind_output(scratch_a.data_ptr(), scratch_i.data_ptr(), H * W)
            ctx.launch(capi.make_frame(W, H, N, 0, path, False, None, (band_h, 1, 0)))
            ctx.sync()
            out["cpu_baseline"]["gpu_frame0_vs_oracle"] = parity_against(ref_acc, ref_win, scratch_a[:H].cpu().numpy())
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
