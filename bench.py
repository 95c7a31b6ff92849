#!/usr/bin/env python3
"""Benchmark of the D-VQVAE grasp-generation path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the whole generation path over one batch of synthetic object
clouds per GPU: GenNet.gen (2 object encoders -> VQ -> autoregressive prior -> codebook
lookup -> decoder -> MANO -> hand encoder -> position decoder), the call-site MANO
(DVQ-VAE/gen_diverse_grasp_obman.py:236-253) and, with several ranks, the RCCL
all-gather of the generated vertices.  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC between the ranks of a node: the host driver here supports no other, and without it RCCL fails with
# "hipIpcGetMemHandle: invalid argument".  The launcher's environment normally carries this already.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd"))
sys.path.insert(0, ROOT)

from dvq_hip import ops, parallel, synth  # noqa: E402
from dvq_hip.pipeline import GraspGenerator  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X
PMC_TRAFFIC_BYTES = 2.173e8                     # encoder trunk, 4096 x 3000 x C4 bf16: FETCH_SIZE x 2 + WRITE_SIZE per launch
PEAK_HBM_GBS = 8000.0                           # HBM3E spec; a bare 16-B-per-lane read of 0.5 GB reaches 5.0-5.4 TB/s here
PEAK_L2_LDS_GBS = 17500.0                       # L2 -> LDS gather rate the part sustains (16.8-18.8 TB/s measured)
FLOP_PER_POINT = {3: 2 * (3 * 64 + 64 * 128 + 128 * 1024), 4: 2 * (4 * 64 + 64 * 128 + 128 * 1024)}
# one hand through the MANO layer in its dense form (DVQ-VAE-2/MANO-master/mano/lbs.py:33-120): PCA 45x45, pose blend
# shapes 135 x 2334, skinning weights x transforms 778 x 16 x 16, transform x vertex 778 x 16, kinematic chain 15 x 4x4x4
MANO_FLOP_PER_HAND = 2 * (45 * 45 + 135 * 2334 + 778 * 16 * 16 + 778 * 16 + 15 * 64)
MANO_BYTES_PER_HAND = 61 * 4 + 778 * 3 * 4 + 16 * 3 * 4


def timed_steps(fn, steps, warmup, device, barrier, kernel_timers=False, finish=None):
    """`finish` completes whatever the last call of `fn` left running (an all-gather in flight); it is
    inside the timed region, so every step's result is complete when the clock stops."""
    for _ in range(warmup):
        fn()
    if finish:
        finish()
    barrier()
    torch.cuda.synchronize(device)
    if kernel_timers:
        ops.enable_timers(True)      # per-launch HIP events over the timed region only
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if finish:
        finish()
    barrier()
    torch.cuda.synchronize(device)
    return time.perf_counter() - t0


def encoder_roofline(N, dtype_name, traffic_note=True):
    """`roofline` object for the object-encoder trunk launches recorded by ops.TIMERS since the last enable_timers."""
    ks = kernel_stats(f"pointnet_mlp_max[C=4,N={N}]")
    if ks is None:
        return None
    avg_ms, launches, (kb, kc, kn), name, alone_ms = ks
    flops = float(kb) * kn * FLOP_PER_POINT[kc]
    ach = flops / avg_ms / 1e9
    peak = PEAK_TFLOPS[dtype_name]
    # HBM bytes per launch from the FETCH_SIZE (x2, the gfx950 correction for 16-B-per-lane streams) and WRITE_SIZE
    # counters: NOT measured by this run -- a constant from separate rocprofv3 --pmc passes over the same kernel and
    # shape (profiles/r3e_pmc_fetch_write.txt: 2.173e8; the pass on another box, r3c: 2.317e8); only valid for that
    # configuration
    traffic = PMC_TRAFFIC_BYTES if (kb, kc, kn, dtype_name) == (4096, 4, 3000, "bf16") else None
    out = {"bound": "mfma", "kernel": name + (" (v_mfma_f32_16x16x32_bf16)" if dtype_name == "bf16" else " (v_mfma_f32_32x32x2_f32)"),
           "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
           "avg_launch_ms": round(avg_ms, 4), "launches_timed": launches, "flop_per_launch": flops}
    if traffic is not None:
        out["traffic_note"] = ("HBM bytes per launch from separate PMC passes (profiles/r3e_pmc_fetch_write.txt; r3c, another "
                               "box: 2.317e8), not from this run; algorithmic bytes 2.134e8")
    return out


def mano_roofline():
    """The MANO layer's launches of the timed steps (two per step: inside gen() and at the call site).  Its arithmetic
    intensity (about 110 FLOP per byte of parameters in and vertices out) is above the fp32 ridge, so the bound it is
    priced against is the fp32 matrix peak; the HBM figure is given beside it."""
    ks = kernel_stats("mano_lbs")
    if ks is None or ks[3] != "mano_lbs":
        return None
    avg_ms, launches, (kb,), name, _ = ks
    flops = float(kb) * MANO_FLOP_PER_HAND
    ach = flops / avg_ms / 1e9
    nbytes = float(kb) * MANO_BYTES_PER_HAND
    return {"bound": "mfma", "kernel": "mano_lbs (v_mfma_f32_16x16x4_f32 pose blend shapes + table skinning)", "achieved": round(ach, 2),
            "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS["fp32"], 4), "traffic": None,
            "avg_launch_ms": round(avg_ms, 5), "launches_timed": launches, "flop_per_launch": flops, "hands_per_launch": kb,
            "hbm_gbs": round(nbytes / avg_ms / 1e6, 1), "frac_hbm_peak": round(nbytes / avg_ms / 1e6 / PEAK_HBM_GBS, 4),
            "note": "dense-form FLOPs (the kernel skips the zero skinning weights: 1-6 of 16 joints per vertex)"}


def kernel_stats(name_prefix):
    """(avg ms, launches, work tuple) over launches recorded by ops.TIMERS."""
    best = None
    for name, recs in (ops.TIMERS or {}).items():
        if name.startswith(name_prefix):
            ms = [a.elapsed_time(b) for a, b, _, _ in recs]
            alone = [a.elapsed_time(b) for a, b, _, main in recs if main]
            cand = (sum(ms) / len(ms), len(ms), recs[0][2], name, (sum(alone) / len(alone)) if alone else None)
            if best is None or cand[0] * cand[1] > best[0] * best[1]:
                best = cand
    return best


def near_ties(z, E, idx, ulps=None):
    """On a sample of rows (z, E as the kernel saw them, widened to fp64): how many have their two nearest codes closer
    than `ulps` fp32 spacings of |z|^2 + max|E|^2 (where the reference's own fp32 rounding decides the winner; default
    max(4, sqrt(D)), the rule of the parity tests), and how many of the kernel's choices are not the exact fp64 nearest
    code / not even within that tolerance of it."""
    if ulps is None:
        ulps = max(4.0, float(E.shape[1]) ** 0.5)
    z64, E64 = z.double(), E.double()
    d = (z64 * z64).sum(1, keepdim=True) + (E64 * E64).sum(1) - 2.0 * z64 @ E64.t()
    top = torch.topk(d, 2, dim=1, largest=False)
    scale = (z.float() ** 2).sum(1) + (E.float() ** 2).sum(1).max()
    tau = ulps * torch.ldexp(torch.ones_like(scale), torch.floor(torch.log2(scale)).to(torch.int32) - 23).double()
    chosen = d.gather(1, idx.view(-1, 1))[:, 0]
    return {"rows": int(z.shape[0]), "near_tie_rows": int((top.values[:, 1] - top.values[:, 0] <= tau).sum()),
            "differs_from_exact_argmin": int((idx != top.indices[:, 0]).sum()),
            "outside_tolerance": int((chosen - top.values[:, 0] > tau).sum())}


def vq_microbench(device, n_rows=1 << 20, iters=5):
    """Isolated nearest-codebook search: 1 Mi latents, reference codebook shapes (128 x 256 and 128 x 1024) + K sweep on
    well-separated latents (distribution B: z = a code + noise), and the reference's own initialisation (distribution A:
    E ~ U(-1/K, 1/K), DVQ-VAE/network/vqvae/quantizer.py:27; z ~ N(0, 0.01^2)) at its codebook shape, with the
    near-tie count of a 4096-row sample."""
    out = []
    for dtype, name in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
        for K, D, dist in ((128, 256, "B"), (512, 256, "B"), (1024, 256, "B"), (2048, 256, "B"), (8192, 256, "B"), (128, 1024, "B"),
                           (128, 256, "A")):
            n = n_rows if D == 256 else n_rows // 4     # the 1024-wide object codebook: same bytes of latents
            g = torch.Generator(device).manual_seed(K)
            if dist == "B":
                E = torch.randn(K, D, device=device, generator=g)
                j = torch.randint(0, K, (n,), device=device, generator=g)
                z = (E[j] + 0.3 * torch.randn(n, D, device=device, generator=g)).to(dtype)
                del j
            else:
                E = (torch.rand(K, D, device=device, generator=g) * 2 - 1) / K
                z = (0.01 * torch.randn(n, D, device=device, generator=g)).to(dtype)
            E = E.to(dtype)
            ws = ops.vq_prepare(E)
            ops.vq_argmin(z, E, want_zq=False, prepared=ws)
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.vq_argmin(z, E, want_zq=False, prepared=ws)
            e1.record()
            torch.cuda.synchronize(device)
            ms = e0.elapsed_time(e1) / iters
            flops = 2.0 * n * K * D
            nbytes = n * D * z.element_size() + K * D * z.element_size() + 8 * n
            rec = {"K": K, "D": D, "dtype": name, "dist": dist, "n": n, "ms": round(ms, 4),
                   "tflops": round(flops / ms / 1e9, 1),
                   "frac_mfma_peak": round(flops / ms / 1e9 / PEAK_TFLOPS[name], 4),
                   "hbm_gbs": round(nbytes / ms / 1e6, 1), "frac_hbm_peak": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4),
                   "mlatents_per_s": round(n / ms / 1e3, 1)}
            if D == 256 and K <= 512:
                idx, _ = ops.vq_argmin(z, E, want_zq=False, prepared=ws)
                rec["near_ties_4096_row_sample"] = near_ties(z[:4096], E, idx[:4096])
                if dist == "B":
                    # the same codebook on latents placed between two codes (gap of the order of the fp32 spacing of
                    # the distances): the rows on which the reference's own rounding decides
                    a = torch.randint(0, K, (4096,), device=device, generator=g)
                    b = (a + 1 + torch.randint(0, K - 1, (4096,), device=device, generator=g)) % K
                    zm = (0.5 * (E[a].double() + E[b].double()) + 1e-5 * torch.randn(4096, D, device=device, generator=g, dtype=torch.float64)).to(dtype)
                    im, _ = ops.vq_argmin(zm, E, want_zq=False, prepared=ws)
                    rec["near_ties_between_two_codes_sample"] = near_ties(zm, E, im)
            out.append(rec)
            del z, E, ws
    return out


def prior_roofline(gen, B, device, iters=20):
    """The prior's dominant launch -- the horizontal-stack gate product of one layer and cell (bf16: [B,2048] x
    [2048,1024] over three taps) -- launched alone with HIP events around it.  It is bound by operand delivery from L2
    into LDS: every workgroup tile (128 rows x 128 or 64 columns) pulls its own weight and activation units."""
    pr = gen.model.GatedPixelCNN
    if pr.compute_dtype != torch.bfloat16 or B < 256:
        return None
    layers, _ = pr._prepared_hip()
    bufs = pr._act_buffers(B, 3, 3, device)
    t, d = layers[1], pr.dim
    kb = d // 32
    taps = [(bufs["XH"][0][1], t.horiz[0], kb), (bufs["XH"][1][1], t.horiz[-1], kb), (bufs["HV"][1][1], t.v2h, 2 * kb)]
    job = ops.prior_job(taps, t.horiz_bias, cond=t.cond, label=bufs["label"], out_packed=bufs["OUT"])
    run = lambda: ops.prior_gemm([job], B, 2 * d, ops.PRIOR_GATE, False, bufs["emb"])
    for _ in range(3):
        run()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    M, N, K = B, 2 * d, 4 * d
    gy = (M + 127) // 128                                  # row tiles of 128 samples
    no = 4 if (N // 128) * gy >= 512 else 2                # column tile of 128 or 64 features (dvq_prior_gemm's rule)
    if no == 2 and (N // 128) * gy >= 256:                 # ... 64-wide tiles pair up in one eight-wave workgroup
        no = 4
    gx = N // (32 * no)
    l2_bytes = 2.0 * K * (gy * N + gx * M)                 # every row tile reads the weights, every column tile the activations
    flops = 2.0 * M * N * K
    ach = l2_bytes / ms / 1e6
    return {"bound": "l2", "kernel": f"prior_gemm gate product [M={M}, N={N}, K={K}], tile 128x{32 * no}",
            "achieved": round(ach, 1), "peak": PEAK_L2_LDS_GBS, "unit": "GB/s", "frac": round(ach / PEAK_L2_LDS_GBS, 4),
            "traffic": None, "avg_launch_ms": round(ms, 5), "launches_timed": iters, "l2_to_lds_bytes_per_launch": l2_bytes,
            "tflops": round(flops / ms / 1e9, 1), "frac_mfma_peak": round(flops / ms / 1e9 / PEAK_TFLOPS["bf16"], 4),
            "note": "launched back to back outside the replayed graph (independent launches: no dependent-chain stalls); "
                    "120 such launches (30 of them, in grid row 0, with one or two taps less) + 120 residual 1x1 products + 30 "
                    "vertical-row launches make one grid batch (grid row 0's vertical stack and cell (0,0) are "
                    "per-class constants of the weights)"}


def cpu_baseline(with_prior, reps=10, batch=16):
    """The CPU oracle (op-for-op restatement of the reference's PyTorch path) on this box's
    host cores, on a bounded sample of the same workload."""
    from oracle import dvq_oracle as O
    sd = synth.gennet_state_dict(0, with_prior=True)
    sd.update(synth.mano_state_dict(0, batch_size=1, prefix="rh_mano."))
    objs = synth.object_clouds(batch, 3000, seed=1234)
    forced = None if with_prior else torch.randint(0, 128, (batch, 3, 3), generator=torch.Generator().manual_seed(0))
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for _ in range(2):                                      # warm-up
            O.generate_grasps(sd, objs[:2], x_tilde=None if forced is None else forced[:2], generator=gen)
        times = []
        t_all = time.perf_counter()
        while len(times) < reps and (len(times) < 3 or time.perf_counter() - t_all < 25.0):
            t0 = time.perf_counter()
            O.generate_grasps(sd, objs, x_tilde=forced, generator=gen)
            times.append(time.perf_counter() - t0)
        el = time.perf_counter() - t_all
        done = batch * len(times)
        med = sorted(times)[len(times) // 2]
        # the reference's literal call pattern: one object per gen() call (a few seconds' worth)
        n1, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 4.0:
            O.generate_grasps(sd, objs[n1 % batch:n1 % batch + 1], x_tilde=None if forced is None else forced[n1 % batch:n1 % batch + 1],
                              generator=gen)
            n1 += 1
        el1 = time.perf_counter() - t1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(batch / med, 3), "unit": "grasps/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model, "value_one_object_per_call": round(n1 / el1, 3),
            "sample": f"median of {len(times)} timed batches of {batch} x 3000-pt clouds ({done} grasps, {el:.1f} s) after 2 warm-up "
                      f"calls, fp32 torch CPU ops, prior {'sampled' if with_prior else 'teacher-forced'}; then {n1} "
                      f"single-object calls, {el1:.1f} s"}


def tta_bench(gen, obj, first_id, device, iters=30):
    """Refinement loop (the step after generation in the reference's adaptation script): time per
    SGD iteration for this batch on the GPU, and the CPU oracle's autograd loop on a small sample."""
    from dvq_hip.tta import GraspRefiner, ITERS
    param = gen.generate(obj, first_id=first_id)["recon_param"]
    r = GraspRefiner(gen.rh_mano)
    r.refine(obj, param, iters=3)                       # builds the buffers and captures the graph
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    r.refine(obj, param, iters=iters)
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) * 1e3 / iters
    B = obj.shape[0]
    out = {"grasps": B, "ms_per_iteration": round(ms, 4), "grasp_iterations_per_s": round(B / ms * 1e3, 1),
           "refined_grasps_per_s_at_300_iterations": round(B / (ms * ITERS) * 1e3, 1)}
    from oracle import tta_oracle as T
    from dvq_hip.hand_parts import CONTACT_VERTS
    sd = synth.mano_state_dict(0)
    nb, ni = 4, 3
    pc, oc = param[:nb].cpu(), obj[:nb, :3].permute(0, 2, 1).contiguous().cpu()
    T.refine(sd, sd["faces_tensor"], oc, pc, CONTACT_VERTS, iters=1)
    t0 = time.perf_counter()
    T.refine(sd, sd["faces_tensor"], oc, pc, CONTACT_VERTS, iters=ni)
    el = time.perf_counter() - t0
    out["cpu_oracle_grasp_iterations_per_s"] = round(nb * ni / el, 2)
    out["cpu_sample"] = f"{nb} grasps x {ni} iterations, torch autograd on {torch.get_num_threads()} threads"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="object clouds per GPU and step")
    ap.add_argument("--points", type=int, default=3000)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--prior", choices=["sampled", "forced"], default="sampled",
                    help="'forced' feeds given part indices instead of running the autoregressive prior")
    ap.add_argument("--tta-iters", type=int, default=0,
                    help="also refine every generated grasp with this many test-time-adaptation steps inside the timed step "
                         "(the reference's adaptation script uses 300); default 0 = generation only")
    ap.add_argument("--no-extras", action="store_true", help="skip the VQ microbench, the variants and the CPU baseline")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: initialise a one-rank process group anyway and issue every collective of the "
                         "multi-rank path (a rehearsal of the RCCL path on a box with one GPU; sets DVQ_FORCE_COLLECTIVE=1)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; 'gloo' rehearses the multi-rank path with several ranks on ONE GPU "
                         "(RCCL refuses two ranks per device)")
    args = ap.parse_args()

    # ONE line on stdout: libraries write there too (RCCL prints a version banner when a communicator is created, gloo its
    # connection notes), so everything but the result line goes to stderr -- file descriptor 1 is pointed at stderr for
    # the run and the result is written to the original one at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    use_dist = world > 1 or args.force_dist
    if args.force_dist:
        os.environ["DVQ_FORCE_COLLECTIVE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if use_dist:
            if args.backend == "nccl":
                dist.barrier(device_ids=[device.index])
            else:
                dist.barrier()

    B, N = args.batch, args.points
    tdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    gen = GraspGenerator.synthetic(device, seed=0, dtype=tdtype, with_prior=True)
    obj = synth.object_clouds_device(B, N, 1234 + rank, device)
    first_id = rank * B
    forced = torch.randint(0, 128, (B, 3, 3), device=device, generator=torch.Generator(device).manual_seed(rank))

    refiner = None
    if args.tta_iters > 0:
        from dvq_hip.tta import GraspRefiner
        refiner = GraspRefiner(gen.rh_mano, iters=args.tta_iters)

    # The all-gather of a step's vertices runs beside the next step's kernels: one step's collective is in
    # flight at a time (GraspGenerator completes the previous step's), and finish_gathers() completes the last one
    # before the clock stops.
    finish_gathers = gen.finish

    def step(prior=args.prior):
        if refiner is None:
            out = gen.generate(obj, first_id=first_id, x_tilde=forced if prior == "forced" else None,
                               gather=use_dist, n_total=world * B, async_gather=not os.environ.get('DVQ_SYNC_GATHER'))
            return out["vertices"]
        # generation, then the adaptation loop on this rank's grasps, then the final hand and the gather
        out = gen.generate(obj, first_id=first_id, x_tilde=forced if prior == "forced" else None)
        p = refiner.refine(obj, out["recon_param"])["recon_param"]
        verts = gen.rh_mano(global_orient=p[:, 10:13].contiguous(), hand_pose=p[:, 13:58].contiguous(),
                            transl=p[:, 58:].contiguous()).vertices
        if use_dist:
            verts = parallel.gather_rows(verts, world * B)
        return verts

    elapsed = timed_steps(step, args.steps, args.warmup, device, barrier, kernel_timers=True, finish=finish_gathers)
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    roof = encoder_roofline(N, args.dtype)
    roof_mano = mano_roofline()
    ops.enable_timers(False)

    # The collective by itself: the same all-gather of one step's vertices with nothing beside it (in the timed
    # steps it runs beside the next step's kernels), max over ranks.
    gather_alone = None
    if use_dist:
        v = torch.randn((B, 778, 3), dtype=torch.float32, device=device)
        el = timed_steps(lambda: parallel.gather_rows(v, world * B), args.steps, 2, device, barrier)
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item()) / args.steps * 1e3
        recv = max(world - 1, 1 if args.force_dist else 0) * B * 778 * 3 * 4
        gather_alone = {"ms": round(ms, 4), "bytes_received_per_rank": recv, "GBs_received_per_rank": round(recv / ms / 1e6, 1),
                        "note": "all-gather of one step's vertices alone, synchronous, nothing else on the device"}
        del v

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        result = {
            "metric": "grasps generated/sec", "value": round(value, 1), "unit": "grasps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "end-to-end GenNet.gen (2 object encoders -> VQ -> prior -> decoder -> MANO -> hand "
                                   "encoder -> pos decoder) + call-site MANO LBS" +
                                   (f" + {'RCCL' if args.backend == 'nccl' else args.backend} all-gather of vertices" if use_dist else "") +
                                   (" (one-rank process group: rehearsal of the multi-rank path)" if args.force_dist and world == 1 else ""),
                       "objects_per_gpu": B, "points_per_object": N, "global_batch": world * B,
                       "prior": args.prior,
                       "prior_note": "every grid is sampled in full each step; what depends on the class label alone (the "
                                     "vertical stack of grid row 0, cell (0,0), and the terms of cells (0,1..) and of grid row 1's "
                                     "vertical stack that read only those) is tabulated per class (128 rows) from the weights, like "
                                     "the packed weights -- no step's inputs or outputs are cached",
                       "tta_iterations": args.tta_iters, "weights": "random-init (seed 0), MANO-shaped synthetic hand model",
                       "parallelism": f"dp{world}"},
        }
        if gather_alone is not None:
            result["all_gather"] = gather_alone
        if roof is not None:
            result["roofline"] = roof
        if roof_mano is not None:
            result["roofline_mano"] = roof_mano

    if not args.no_extras and world == 1 and not args.force_dist:
        # variants of the same step, the VQ microbench and the CPU baseline (rank 0, N=1 only)
        variants = {}
        other = "forced" if args.prior == "sampled" else "sampled"
        el = timed_steps(lambda: step(other), max(2, args.steps // 2), 1, device, barrier)
        variants[f"{args.dtype}_prior_{other}"] = round(B * max(2, args.steps // 2) / el, 1)
        odt = torch.float32 if args.dtype == "bf16" else torch.bfloat16
        import dvq_hip
        dvq_hip.set_compute_dtype(gen.model, odt)
        el = timed_steps(lambda: step("forced"), 2, 1, device, barrier)
        oname = 'fp32' if args.dtype == 'bf16' else 'bf16'
        variants[f"{oname}_prior_forced"] = round(B * 2 / el, 1)
        # the other precision as a leg of its own: the same step, the same number of timed steps and warm-up steps, its
        # own per-launch events (the reference's arithmetic is fp32; the headline follows the configuration's bf16)
        el = timed_steps(lambda: step("sampled"), args.steps, args.warmup, device, barrier, kernel_timers=True)
        roof_o = encoder_roofline(N, oname)
        ops.enable_timers(False)
        variants[f"{oname}_prior_sampled"] = round(B * args.steps / el, 1)
        result[f"{oname}_leg"] = {"value": round(B * args.steps / el, 1), "unit": "grasps/s", "ms_per_step": round(el / args.steps * 1e3, 3),
                                  "steps": args.steps, "warmup": args.warmup, "dtype": oname, "prior": "sampled"}
        if roof_o is not None:
            result[f"roofline_{oname}"] = roof_o
        dvq_hip.set_compute_dtype(gen.model, tdtype)
        # The random-init object codebook sends every synthetic cloud to ONE class, the best case for the prior's
        # per-class table reads.  Trained-like case: the object codebook set to the features of 128 of the batch's own
        # clouds, so that the batch spreads over (up to) 128 classes; restored afterwards.
        if args.prior == "sampled":
            E6 = gen.model.vqvae6.vector_quantization.embedding.weight
            saved = E6.detach().clone()
            with torch.no_grad():
                feat = gen.model.obj_encoder_type(obj)[0]
                pick = torch.randperm(B, device=device, generator=torch.Generator(device).manual_seed(7))[:E6.shape[0]]
                E6.copy_(feat[pick].to(E6.dtype))
            el = timed_steps(lambda: step("sampled"), args.steps, 2, device, barrier)
            variants[f"{args.dtype}_prior_sampled_diverse_classes"] = round(B * args.steps / el, 1)
            variants["diverse_classes_in_batch"] = int(gen.model.last_idx6.unique().numel())
            with torch.no_grad():
                E6.copy_(saved)
        result["variants_grasps_per_s"] = variants
        pr = prior_roofline(gen, B, device)
        if pr is not None:
            result["roofline_prior"] = pr
        result["tta_refinement"] = tta_bench(gen, obj, first_id, device)
        del obj
        torch.cuda.empty_cache()
        result["vq_argmin"] = vq_microbench(device)
        result["cpu_baseline"] = cpu_baseline(args.prior == "sampled")

    if rank == 0:
        sys.stdout.flush()
        with os.fdopen(result_fd, "w") as out:
            out.write(json.dumps(result) + "\n")
    if use_dist:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
