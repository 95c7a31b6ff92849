#!/usr/bin/env python3
"""Time the reference's own GenNet.gen against the CPU oracle on the same cores (build container only).

    python tools/cpu_ref_vs_oracle.py [--reps 12] [--threads 8]

bench.py's `cpu_baseline` times the oracle (kind "port") because the reference's Python cannot travel to the GPU box.
This script backs that choice with a same-box ratio: it imports the reference's modules exactly as
tests/golden/make_golden.py does (same shims, same seeded weights, strict load_state_dict), and times, alternating
between the two, one object per call (the reference's only valid batch size, quantizer.py:70-74):
  * reference: model.gen(obj[b:b+1]) + the call-site parameter assembly and MANO (gen_diverse_grasp_obman.py:236-253)
  * oracle:    oracle.dvq_oracle.generate_grasps(sd, obj[b:b+1])
Both sample the prior (nine full forwards of the 15-layer PixelCNN, pixelcnn/models.py:176-198).  Also times the
isolated quantizer forward on 1 Mi x 256 latents against 128 codes (quantizer.py:44-54) both ways.
Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    import make_golden as G                      # puts d-vqvae_amd on the path and imports dvq_hip.synth
    from dvq_hip import synth
    from oracle import dvq_oracle as O
    ref = G.import_reference()
    torch.set_num_threads(args.threads)

    sd = synth.gennet_state_dict(0, with_prior=True)
    model = ref.gn.GenNet()
    model.load_state_dict(sd)
    model.eval()
    rh = G.ref_mano(ref, 0, batch_size=1).eval()
    model.set_rh_mano(rh)
    osd = dict(sd)
    osd.update(synth.mano_state_dict(0, batch_size=1, prefix="rh_mano."))
    objs = synth.object_clouds(args.reps + 2, 3000, seed=1234)

    def run_reference(b):
        with G.cuda_to_cpu(), torch.no_grad():
            r, p = model.gen(objs[b:b + 1])
            prm = torch.zeros(1, 61)
            prm[:, 0:10] = r[:, 0:10]; prm[:, 10:13] = p[:, 0:3]; prm[:, 13:58] = r[:, 10:55]; prm[:, 58:61] = p[:, 3:6]
            return rh(betas=prm[:, :10], global_orient=prm[:, 10:13], hand_pose=prm[:, 13:58], transl=prm[:, 58:]).vertices

    gen = torch.Generator().manual_seed(0)

    def run_oracle(b):
        with torch.no_grad():
            return O.generate_grasps(osd, objs[b:b + 1], generator=gen)["vertices"]

    for b in range(2):                            # warm-up, both
        run_reference(b)
        run_oracle(b)
    t_ref, t_or = [], []
    for b in range(2, args.reps + 2):             # alternate so that drift of the box hits both alike
        t0 = time.perf_counter(); run_reference(b); t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); run_oracle(b); t_or.append(time.perf_counter() - t0)

    # isolated quantizer: the reference module and the oracle's restatement, 1 Mi latents of the reference's shape
    z, E = synth.vq_case(128, 256, 1 << 20, "B", seed=0)
    vq = ref.quantizer.VectorQuantizer(128, 256, 0.25, 1)
    vq.embedding.weight.data.copy_(E)
    tq_ref, tq_or = [], []
    with torch.no_grad():
        vq(z[:4096], False); O.vq_forward(z[:4096], E)
        for _ in range(3):
            t0 = time.perf_counter(); vq(z, False); tq_ref.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); O.vq_forward(z, E); tq_or.append(time.perf_counter() - t0)

    out = {"threads": args.threads, "reps": args.reps,
           "gen_one_object_per_call_s": {"reference": round(median(t_ref), 4), "oracle": round(median(t_or), 4)},
           "gen_grasps_per_s": {"reference": round(1 / median(t_ref), 3), "oracle": round(1 / median(t_or), 3)},
           "oracle_over_reference_time": round(median(t_or) / median(t_ref), 3),
           "vq_1Mi_x_256_vs_128_codes_s": {"reference": round(median(tq_ref), 3), "oracle": round(median(tq_or), 3)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
