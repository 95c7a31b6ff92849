#!/usr/bin/env python3
"""Generate diverse grasps for object meshes/clouds on one MI355X and write the reference's JSON.

The model part of DVQ-VAE/gen_diverse_grasp_ho3d.py (load checkpoints :359-384, per object
--num_grasp random placements -> model.gen -> 61-d parameters -> MANO :212-257, one JSON per
object :322-331) with all placements of an object in one batch.  The reference's grasp-quality
evaluation around it (trimesh/pybullet/V-HACD, :252-313) is not part of this path.

    python tools/gen_diverse_grasp.py --objects a.obj b.npy --out diverse_grasp/ho3d \
        --checkpoint checkpoints/model_best.pth --prior checkpoints/LATENT_BLOCK_pixelcnn.pt \
        --mano models/mano/MANO_RIGHT.pkl
Without --checkpoint the seeded synthetic weights are used (smoke runs).
An object is a Wavefront .obj mesh, or a .npy array of >=3000 surface points [*,3]
(the reference's `resampled.npy`; its scale is then the diagonal of the points' bounding box).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd"))


def build_generator(args, device):
    import dvq_hip
    from dvq_hip.pipeline import GraspGenerator
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if not args.checkpoint:
        return GraspGenerator.synthetic(device, seed=args.seed, dtype=dtype)
    import mano
    from network.gen_net import GenNet
    model = GenNet()
    ckpt = torch.load(args.checkpoint, map_location="cpu", weights_only=True)["network"]
    sd = model.state_dict()
    sd.update({k: v for k, v in ckpt.items() if k in sd})
    model.load_state_dict(sd)
    model.GatedPixelCNN.load_state_dict(torch.load(args.prior, map_location="cpu", weights_only=True))
    model = model.to(device).eval()
    rh = mano.load(model_path=args.mano, model_type="mano", use_pca=True, num_pca_comps=45, batch_size=1,
                   flat_hand_mean=True).to(device).eval()
    model.set_rh_mano(rh)
    dvq_hip.set_compute_dtype(model, dtype)
    return GraspGenerator(model, rh, seed=args.seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", nargs="+", required=True)
    ap.add_argument("--out", default="./diverse_grasp/ho3d")
    ap.add_argument("--num_grasp", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", choices=("bf16", "fp32"), default="fp32")
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--prior", default="./checkpoints/LATENT_BLOCK_pixelcnn.pt")
    ap.add_argument("--mano", default="./models/mano/MANO_RIGHT.pkl")
    ap.add_argument("--tta-iters", type=int, default=0,
                    help="adapt every generated grasp to its object with this many SGD steps before it is written "
                         "(DVQ-VAE/gen_HDMO_TTA.py:119-152 uses 300); 0 = generation only")
    ap.add_argument("--closed-faces", default="",
                    help="closed hand mesh for the adaptation's vertex normals (the reference reads "
                         "./models/mano/closed_mano_faces.pkl, gen_HDMO_TTA.py:61-62); default: the model's own faces")
    args = ap.parse_args()

    from dvq_hip import grasp_io
    device = torch.device("cuda", 0)
    gen = build_generator(args, device)
    refiner = None
    if args.tta_iters > 0:
        from dvq_hip.tta import GraspRefiner, read_closed_faces
        faces = read_closed_faces(args.closed_faces) if args.closed_faces else None
        refiner = GraspRefiner(gen.rh_mano, faces=faces, iters=args.tta_iters)
    rs = np.random.RandomState(args.seed)
    for k, path in enumerate(args.objects):
        if path.endswith(".npy"):
            pts = np.load(path, allow_pickle=False)
            obj_pc = grasp_io.make_obj_pc(pts, grasp_io.get_diameter(pts))
        else:
            obj_pc, _, _ = grasp_io.obj_pc_from_mesh(path, seed=args.seed)
        record, verts = grasp_io.diverse_grasps(gen, obj_pc, args.num_grasp, rs, first_id=k * args.num_grasp, device=device,
                                                refiner=refiner)
        name = os.path.splitext(os.path.basename(path))[0]
        digits = "".join(ch for ch in name[:3] if ch.isdigit())
        out = os.path.join(args.out, f"obj_id_{int(digits)}.json" if len(digits) == 3 else f"obj_{name}.json")
        grasp_io.write_grasp_json(out, record)
        print(f"{path}: {verts.shape[0]} grasps -> {out}", flush=True)


if __name__ == "__main__":
    main()
