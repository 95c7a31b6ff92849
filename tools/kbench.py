#!/usr/bin/env python3
"""Kernel-level timings for tuning: python tools/kbench.py [pointnet] [vq] [mano] [linear]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd"))
from dvq_hip import ops, synth  # noqa: E402
from dvq_hip._lib import DVQ_BF16, DVQ_F32  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def pointnet():
    from network.pointnet_encoder import PointNetEncoder
    sd = synth.gennet_state_dict(0, with_prior=False)
    for C, N, B in [tuple(int(v) for v in os.environ["DVQ_PN_CASES"].split(","))] if os.environ.get("DVQ_PN_CASES") else ((4, 3000, 4096), (3, 778, 4096), (3, 212, 4096), (3, 123, 4096), (3, 83, 4096), (4, 3000, 64), (4, 3000, 1)):
        enc = PointNetEncoder(channel=C)
        pre = "obj_encoder_type." if C == 4 else "recon_encoder."
        enc.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
        enc = enc.to(dev).eval()
        x = torch.randn(B, C, N, device=dev) * 0.1
        for dt, code, peak in ((torch.bfloat16, DVQ_BF16, 2500.0), (torch.float32, DVQ_F32, 157.3)):
            enc.compute_dtype = dt
            w = enc._packed().as_dict()
            trans = torch.eye(3, device=dev).repeat(B, 1, 1).contiguous()
            ms = timeit(lambda: ops.pointnet_mlp_max(x, trans, w, False, code), iters=5 if dt == torch.float32 else 10)
            fl = 2.0 * B * N * (C * 64 + 64 * 128 + 128 * 1024)
            print(f"pointnet_mlp_max C={C} N={N} B={B} {str(dt)[6:]:9s}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  "
                  f"{fl / ms / 1e9 / peak * 100:5.1f}% of peak", flush=True)


def vq():
    n = 1 << 20
    for dt, name, peak in ((torch.bfloat16, "bf16", 2500.0), (torch.float32, "fp32", 157.3)):
        for K, D in ((128, 256), (256, 256), (512, 256), (1024, 256), (2048, 256), (8192, 256), (128, 1024)):
            nn = n if D == 256 else n // 4
            g = torch.Generator(dev).manual_seed(K)
            E = torch.randn(K, D, device=dev, generator=g)
            z = (E[torch.randint(0, K, (nn,), device=dev, generator=g)] + 0.3 * torch.randn(nn, D, device=dev, generator=g)).to(dt)
            E = E.to(dt)
            ws = ops.vq_prepare(E)
            ms = timeit(lambda: ops.vq_argmin(z, E, want_zq=False, prepared=ws), iters=5)
            fl = 2.0 * nn * K * D
            by = nn * D * z.element_size() + 8 * nn
            print(f"vq_argmin K={K:5d} D={D:4d} n={nn} {name}: {ms:8.4f} ms  {fl / ms / 1e9:7.1f} TFLOP/s {fl / ms / 1e9 / peak * 100:5.1f}% of peak"
                  f"  {by / ms / 1e6:7.1f} GB/s", flush=True)
            del z, E, ws


def vq_stream():
    """How fast the latent stream of the reference codebook shape (128 x 256, bf16) goes as the batch grows, next to
    plain reads of the same bytes (torch sum of the buffer as int32, and a device-to-device copy)."""
    K, D = 128, 256
    g = torch.Generator(dev).manual_seed(1)
    E = torch.randn(K, D, device=dev, generator=g).to(torch.bfloat16)
    ws = ops.vq_prepare(E)
    for n in (1 << 18, 1 << 20, 1 << 22, 1 << 23):
        z = torch.randn(n, D, device=dev, generator=g).to(torch.bfloat16)
        idx = torch.empty(n, dtype=torch.int64, device=dev)
        ms = timeit(lambda: ops.vq_argmin(z, E, want_zq=False, prepared=ws), iters=10)
        by = n * D * 2 + 8 * n
        zi = z.view(torch.int32)
        ms_sum = timeit(lambda: zi.sum(), iters=10)
        out = torch.empty_like(z)
        ms_cp = timeit(lambda: out.copy_(z), iters=10)
        print(f"n={n:8d}: vq_argmin {ms:7.4f} ms {by / ms / 1e6:7.1f} GB/s | int32 sum {ms_sum:7.4f} ms {n * D * 2 / ms_sum / 1e6:7.1f} GB/s"
              f" | copy {ms_cp:7.4f} ms {2 * n * D * 2 / ms_cp / 1e6:7.1f} GB/s (read+write)", flush=True)
        del z, out, zi, idx


def vq_obj():
    """The object codebook search as gen() runs it: one 1024-d row per object, 128 codes, fp32."""
    for n in (4096, 256, 1):
        g = torch.Generator(dev).manual_seed(1)
        E = torch.randn(128, 1024, device=dev, generator=g)
        z = torch.randn(n, 1024, device=dev, generator=g)
        ws = ops.vq_prepare(E)
        ms = timeit(lambda: ops.vq_argmin(z, E, want_zq=False, prepared=ws), iters=10)
        print(f"vq_argmin K=128 D=1024 n={n} fp32: {ms * 1e3:8.1f} us", flush=True)


def mano():
    import mano as M
    from dvq_hip.mano_io import Struct
    rh = M.MANO('', data_struct=Struct(**synth.mano_model_data(0)), use_pca=True, num_pca_comps=45, batch_size=1,
                flat_hand_mean=True).to(dev).eval()
    for B in (1, 64, 4096, 32768):
        go, hp, tr = (t.to(dev) for t in synth.mano_inputs(B, seed=1))
        ms = timeit(lambda: rh(global_orient=go, hand_pose=hp, transl=tr))
        print(f"mano B={B}: {ms:.4f} ms  {B * 778 * 12 / ms / 1e6:.1f} GB/s out", flush=True)


def linear():
    for M_, K, N in ((4096, 2560, 1024), (4096, 1024, 256), (4096, 1024, 512), (4096, 2048, 1024), (1, 2560, 1024), (16, 2560, 1024)):
        x = torch.randn(M_, K, device=dev)
        W = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        ms = timeit(lambda: ops.linear(x, W, b, relu=True))
        print(f"linear M={M_} K={K} N={N}: {ms:.4f} ms  {2.0 * M_ * K * N / ms / 1e9:.1f} TFLOP/s", flush=True)
        if M_ >= ops.LINEAR_BF16_MIN_ROWS:
            Wb = ops.to_bf16(W)
            ms = timeit(lambda: ops.linear(x, W, b, relu=True, W_bf16=Wb))
            print(f"linear M={M_} K={K} N={N} bf16 operands: {ms:.4f} ms  {2.0 * M_ * K * N / ms / 1e9:.1f} TFLOP/s", flush=True)


def heads():
    """An MLP head's two wide layers (decoder: 2560 -> 1024 -> 256, 4096 rows, bf16 operands): the register-operand
    dvq_linear_bf16 kernel against the LDS-staged product kernel of the prior on packed activations."""
    M_ = 4096
    g = torch.Generator(dev).manual_seed(0)
    pieces = [torch.randn(M_, 256, device=dev, generator=g) for _ in range(6)] + [torch.randn(M_, 1024, device=dev, generator=g)]
    x = torch.cat(pieces, 1).contiguous()
    W0 = torch.randn(1024, 2560, device=dev, generator=g) * 0.02
    b0 = torch.randn(1024, device=dev, generator=g) * 0.1
    W1 = torch.randn(256, 1024, device=dev, generator=g) * 0.03
    b1 = torch.randn(256, device=dev, generator=g) * 0.1
    W0b, W1b = ops.to_bf16(W0), ops.to_bf16(W1)

    def old():
        h = ops.linear(x, W0, b0, relu=True, W_bf16=W0b)
        return ops.linear(h, W1, b1, relu=True, W_bf16=W1b)

    offs = [0, 256, 512, 768, 1024, 1280, 1536, 2560]
    W0p = [ops.prior_pack_weight(W0[:, a:b]) for a, b in zip(offs[:-1], offs[1:])]
    W1p = ops.prior_pack_weight(W1)
    ar = torch.arange(M_, device=dev)

    def new():
        taps = []
        for pc, wp in zip(pieces, W0p):
            buf = ops.prior_packed_empty(M_, pc.shape[1], dev)
            ops.prior_embed(ar, pc, buf)
            taps.append((buf, wp, pc.shape[1] // 32))
        h = ops.prior_packed_empty(M_, 1024, dev)
        ops.prior_gemm([ops.prior_job(taps, b0, out_packed=h)], M_, 1024, ops.PRIOR_PACK, True, b0)
        out = torch.empty(M_, 256, device=dev)
        ops.prior_gemm([ops.prior_job([(h, W1p, 32)], b1, out_rows=out)], M_, 256, ops.PRIOR_ROWS, True, b0)
        return out

    a, b = old(), new()
    ref = torch.relu(torch.relu(x.to(torch.bfloat16).float() @ W0.to(torch.bfloat16).float().t() + b0).to(torch.bfloat16).float()
                     @ W1.to(torch.bfloat16).float().t() + b1)
    print(f"heads: old vs fp32-of-bf16 reference {(a - ref).abs().max().item():.3e}, new {(b - ref).abs().max().item():.3e}, "
          f"scale {ref.abs().max().item():.3f}", flush=True)
    print(f"heads old: {timeit(old):.4f} ms   new (7 embeds + 2 products): {timeit(new):.4f} ms", flush=True)


def prior():
    from network.gen_net import GenNet
    from dvq_hip import rng
    model = GenNet()
    model.load_state_dict(synth.gennet_state_dict(0))
    pr = model.GatedPixelCNN.to(dev).eval()
    for B in (4096, 2048, 512, 1):
        lab = torch.randint(0, 128, (B,), device=dev)
        u = rng.uniforms(torch.arange(B, device=dev), 9, 0)
        for dt in (torch.bfloat16, torch.float32):
            pr.compute_dtype = dt
            ms = timeit(lambda: pr.generate(None, lab, shape=(3, 3), batch_size=B, uniforms=u), iters=3, warm=1)
            print(f"prior.generate B={B} {str(dt)[6:]:9s}: {ms:8.3f} ms  ({1.22e9 * B / ms / 1e9:7.1f} TFLOP/s useful)", flush=True)


def prior_classes():
    """prior.generate with grid row 0 / cell (0,0) evaluated once per class against once per row."""
    from network.gen_net import GenNet
    from dvq_hip import rng
    model = GenNet()
    model.load_state_dict(synth.gennet_state_dict(0))
    pr = model.GatedPixelCNN.to(dev).eval()
    sizes = [int(v) for v in os.environ.get("DVQ_SIZES", "4096,2048,1024,512,256").split(",")]
    for dt in (torch.bfloat16, torch.float32):
        pr.compute_dtype = dt
        pr.class_rows_min_rows = 1
        for B in sizes:
            lab = torch.randint(0, 128, (B,), device=dev)
            u = rng.uniforms(torch.arange(B, device=dev), 9, 0)
            res = {}
            for on in (False, True, False, True):
                pr.class_rows = on
                ms = timeit(lambda: pr.generate(None, lab, shape=(3, 3), batch_size=B, uniforms=u), iters=5, warm=2)
                res[on] = pr.generate(None, lab, shape=(3, 3), batch_size=B, uniforms=u)
                print(f"prior.generate B={B} {str(dt)[6:]:9s} per class={int(on)}: {ms:8.3f} ms", flush=True)
            print(f"   grids equal: {(res[True] == res[False]).float().mean().item():.6f}", flush=True)


def prior_split():
    """prior.generate at 4096 rows with the batch cut into sub-batches on concurrent streams inside the graph."""
    from network.gen_net import GenNet
    from dvq_hip import rng
    model = GenNet()
    model.load_state_dict(synth.gennet_state_dict(0))
    pr = model.GatedPixelCNN.to(dev).eval()
    pr.compute_dtype = torch.bfloat16
    B = 4096
    lab = torch.randint(0, 128, (B,), device=dev)
    u = rng.uniforms(torch.arange(B, device=dev), 9, 0)
    ref = None
    cfgs = ((1, 3072), (2, 3072), (2, 1024), (3, 1024), (4, 1024), (4, 512), (1, 3072))
    if os.environ.get("DVQ_SPLIT"):
        cfgs = ((int(os.environ["DVQ_SPLIT"]), 1024),)
    for split, vmin in cfgs:
        pr.split, pr.vrow_min_rows = split, vmin
        ms = timeit(lambda: pr.generate(None, lab, shape=(3, 3), batch_size=B, uniforms=u), iters=5, warm=2)
        x = pr.generate(None, lab, shape=(3, 3), batch_size=B, uniforms=u)
        ref = x if ref is None else ref
        print(f"prior.generate B={B} bf16 split={split} vrow_min={vmin}: {ms:8.3f} ms; cells equal to unsplit {(x == ref).float().mean().item():.4f}",
              flush=True)


def dvqvae():
    """The encode path (DVQVAE.forward, eval): two object encoders, six hand-part encoders + MLPs, seven codebook searches."""
    from network.DVQVAE import DVQVAE
    model = DVQVAE()
    model.load_state_dict(synth.dvqvae_state_dict(0))
    model = model.to(dev).eval()
    for B in (4096, 64):
        obj = synth.object_clouds_device(B, 3000, 7, dev)
        hand = torch.randn(B, 3, 778, device=dev) * 0.05
        for dt in (torch.bfloat16, torch.float32):
            import dvq_hip
            dvq_hip.set_compute_dtype(model, dt)
            with torch.no_grad():
                ms = timeit(lambda: model(obj, hand), iters=3, warm=1)
            print(f"DVQVAE.forward(eval) B={B} {str(dt)[6:]:9s}: {ms:8.3f} ms  {B / ms * 1e3:10.0f} hands/s", flush=True)


def gen():
    """End-to-end generation (gen + call-site MANO) at small batch sizes: the reference calls it with one object at a time."""
    from dvq_hip.pipeline import GraspGenerator
    for dt in (torch.bfloat16, torch.float32):
        g = GraspGenerator.synthetic(dev, seed=0, dtype=dt)
        for B in (1, 16, 256):
            obj = synth.object_clouds_device(B, 3000, 11, dev)
            ms = timeit(lambda: g.generate(obj), iters=5, warm=2)
            print(f"generate B={B} {str(dt)[6:]:9s}: {ms:8.3f} ms  {B / ms * 1e3:10.0f} grasps/s", flush=True)


def gen4096():
    """The bench step (4096 objects x 3000 points, bf16), three timings of ten steps."""
    from dvq_hip.pipeline import GraspGenerator
    g = GraspGenerator.synthetic(dev, seed=0, dtype=torch.bfloat16)
    obj = synth.object_clouds_device(4096, 3000, 11, dev)
    ts = [timeit(lambda: g.generate(obj), iters=10, warm=3) for _ in range(3)]
    print(f"generate B=4096 bf16: {min(ts):8.3f} .. {max(ts):8.3f} ms  {4096 / min(ts) * 1e3:10.0f} grasps/s", flush=True)


def tta():
    import mano as M
    from dvq_hip.mano_io import Struct
    from dvq_hip.tta import GraspRefiner
    rh = M.MANO('', data_struct=Struct(**synth.mano_model_data(0)), use_pca=True, num_pca_comps=45, batch_size=1,
                flat_hand_mean=True).to(dev).eval()
    for B in (4096, 256, 1):
        go, hp, tr = (t.to(dev) for t in synth.mano_inputs(B, seed=1))
        param = torch.zeros(B, 61, device=dev)
        param[:, 10:13], param[:, 13:58], param[:, 58:] = go, hp, tr
        verts = rh(global_orient=go, hand_pose=hp, transl=tr).vertices
        obj = verts.mean(1, keepdim=True) + 0.04 * torch.randn(B, 3000, 3, device=dev)
        obj_pc = torch.cat([obj.permute(0, 2, 1), torch.full((B, 1, 3000), 0.2, device=dev)], 1).contiguous()
        r = GraspRefiner(rh)
        iters = 50
        ms = timeit(lambda: r.refine(obj_pc, param, iters=iters), iters=2, warm=1)
        print(f"tta refine B={B}: {ms / iters:8.4f} ms/iteration  {B * iters / ms * 1e3:10.0f} grasp-iterations/s  "
              f"(300 iterations: {ms / iters * 300:8.1f} ms per batch)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["pointnet", "vq", "mano", "linear"]
    for w in which:
        globals()[w]()
