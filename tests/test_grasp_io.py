"""Host-side formats around the generation path: object clouds in, the reference's JSON out."""
import json
import os

import numpy as np
import pytest
import torch

from dvq_hip import grasp_io


def _cube():
    v = np.array([[x, y, z] for x in (0, 1) for y in (0, 2) for z in (0, 3)], dtype=np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    return v, quads


def test_diameter_and_obj_pc_layout():
    g = np.random.default_rng(0)
    pts = g.normal(size=(3500, 3))
    ext = pts.max(0) - pts.min(0)
    assert grasp_io.get_diameter(pts) == pytest.approx(np.sqrt((ext ** 2).sum()), rel=1e-12)
    pc = grasp_io.make_obj_pc(pts, 0.25)
    assert pc.shape == (4, 3000) and pc.dtype == torch.float32 and pc.is_contiguous()
    assert torch.equal(pc[:3].T, torch.tensor(pts[:3000], dtype=torch.float32))
    assert torch.all(pc[3] == torch.tensor(0.25))
    with pytest.raises(ValueError):
        grasp_io.make_obj_pc(pts[:100], 0.25)


def test_placements_follow_the_reference_stream():
    # the reference draws 3 angles then 3 (zero-weighted) jitter values per grasp from numpy's global stream
    np.random.seed(7)
    raw = np.random.random(12)
    np.random.seed(7)
    pl = grasp_io.random_placements(2)
    assert np.array_equal(pl["r"][0], raw[0:3] * np.pi * 2)
    assert np.array_equal(pl["r"][1], raw[6:9] * np.pi * 2)
    assert np.array_equal(pl["trans"][1].ravel(), grasp_io.DEFAULT_TRANSLATION)
    tx, ty, tz = pl["r"][0]
    rx = np.array([[1, 0, 0], [0, np.cos(tx), -np.sin(tx)], [0, np.sin(tx), np.cos(tx)]])
    ry = np.array([[np.cos(ty), 0, np.sin(ty)], [0, 1, 0], [-np.sin(ty), 0, np.cos(ty)]])
    rz = np.array([[np.cos(tz), -np.sin(tz), 0], [np.sin(tz), np.cos(tz), 0], [0, 0, 1]])
    assert np.array_equal(pl["R"][0][:, :3], rx @ ry @ rz)
    assert np.allclose(pl["R"][0][:, :3] @ pl["R"][0][:, :3].T, np.eye(3), atol=1e-12)
    assert pl["R"].shape == (2, 3, 4) and np.array_equal(pl["R"][1][:, 3:], pl["trans"][1])


def test_place_clouds_matches_the_per_grasp_formula():
    g = np.random.default_rng(1)
    pc = grasp_io.make_obj_pc(g.normal(size=(3000, 3)) * 0.05, 0.2)
    pl = grasp_io.random_placements(3, np.random.RandomState(3))
    out = grasp_io.place_clouds(pc, pl["R"])
    assert out.shape == (3, 4, 3000)
    xyz = pc.permute(1, 0)[:, :3].numpy()                                   # as the reference takes it
    for i in range(3):
        R = pl["R"][i]
        ref = np.matmul(R[:3, 0:3], xyz.copy().T) + R[:3, 3].reshape(-1, 1)
        assert torch.equal(out[i, :3], torch.tensor(ref, dtype=torch.float32))
        assert torch.equal(out[i, 3], pc[3])


def test_json_record_is_what_the_reference_scripts_exchange(tmp_path):
    n = 5
    pl = grasp_io.random_placements(n, np.random.RandomState(0))
    params = torch.randn(n, 61)
    rec = grasp_io.grasp_record(params, pl)
    assert set(rec) == {"recon_params", "R_list", "trans_list", "r_list"}
    path = os.path.join(tmp_path, "sub", "obj_id_3.json")
    grasp_io.write_grasp_json(path, rec)
    with open(path) as f:
        raw = json.load(f)
    # diverse_grasp/diversity.py reads data['recon_params'][i][0] as one 61-vector
    assert len(raw["recon_params"]) == n and len(raw["recon_params"][2]) == 1 and len(raw["recon_params"][2][0]) == 61
    assert np.asarray(raw["R_list"]).shape == (n, 3, 4)
    assert np.asarray(raw["trans_list"]).shape == (n, 3, 1)
    assert np.asarray(raw["r_list"]).shape == (n, 3)
    back = grasp_io.read_grasp_json(path)
    assert np.array_equal(back["recon_params"], params.numpy())
    assert np.array_equal(back["R"], pl["R"]) and np.array_equal(back["r"], pl["r"])
    with pytest.raises(ValueError):
        grasp_io.grasp_record(params[:2], pl)


def test_obj_reader_and_surface_sampler(tmp_path):
    v, quads = _cube()
    path = os.path.join(tmp_path, "textured_simple.obj")
    with open(path, "w") as f:
        f.write("# cube\n")
        for p in v:
            f.write(f"v {p[0]} {p[1]} {p[2]}\n")
        f.write("vt 0 0\n")
        for q in quads:
            f.write("f " + " ".join(f"{i + 1}/1" for i in q) + "\n")
    verts, faces = grasp_io.load_obj(path)
    assert verts.shape == (8, 3) and faces.shape == (12, 3) and faces.min() == 0 and faces.max() == 7
    pts, fid = grasp_io.sample_surface(verts, faces, 20000, seed=1)
    assert pts.shape == (20000, 3) and fid.shape == (20000,)
    lo, hi = np.zeros(3), np.array([1.0, 2.0, 3.0])
    assert (pts >= lo - 1e-12).all() and (pts <= hi + 1e-12).all()
    on_face = (np.isclose(pts, lo, atol=1e-9) | np.isclose(pts, hi, atol=1e-9)).any(axis=1)
    assert on_face.all()
    # area-proportional: the two 2x3 faces (x = 0, x = 1) carry 12/22 of the surface
    frac = np.mean(np.isclose(pts[:, 0], 0, atol=1e-9) | np.isclose(pts[:, 0], 1, atol=1e-9))
    assert abs(frac - 12 / 22) < 0.02
    pc, v2, f2 = grasp_io.obj_pc_from_mesh(path, seed=0)
    assert pc.shape == (4, 3000) and float(pc[3, 0]) == pytest.approx(np.sqrt(1 + 4 + 9), rel=1e-6)


def test_diversity_metric():
    g = np.random.default_rng(0)
    centres = g.normal(size=(4, 61)) * 5
    x = np.concatenate([c + 0.01 * g.normal(size=(50, 61)) for c in centres])
    ent, dist = grasp_io.diversity(x, cls_num=4, seed=0)
    assert 0.0 <= ent <= np.log(4) + 1e-9 and dist < 1.0


@pytest.mark.gpu
def test_diverse_grasps_batch_equals_one_call_per_placement():
    """All placements of an object in one batch = the reference's loop of single calls."""
    from dvq_hip.pipeline import GraspGenerator
    dev = torch.device("cuda", 0)
    gen = GraspGenerator.synthetic(dev, seed=0, dtype=torch.float32)
    g = np.random.default_rng(2)
    pc = grasp_io.make_obj_pc(g.uniform(-0.05, 0.05, size=(3000, 3)), 0.17)
    rec, verts = grasp_io.diverse_grasps(gen, pc, n_grasps=6, rng=np.random.RandomState(5), first_id=40, device=dev)
    assert verts.shape == (6, 778, 3) and len(rec["recon_params"]) == 6
    pl = grasp_io.random_placements(6, np.random.RandomState(5))
    clouds = grasp_io.place_clouds(pc, pl["R"]).to(dev)
    for i in (0, 3, 5):
        one = gen.generate(clouds[i:i + 1], first_id=40 + i)
        # batch-of-6 and batch-of-1 take different matrix-product code paths (summation order): float32 noise only
        assert torch.allclose(one["vertices"][0], verts[i], rtol=0, atol=2e-6), float((one["vertices"][0] - verts[i]).abs().max())
        assert np.allclose(np.asarray(rec["recon_params"][i][0], dtype=np.float32), one["recon_param"][0].cpu().numpy(), atol=2e-5)


@pytest.mark.gpu
def test_diverse_grasps_with_adaptation_records_the_adapted_grasps():
    """With a refiner the recorded parameters are the generated ones after the adaptation steps
    (gen_HDMO_TTA.py:113-163) and the vertices returned are the hand of the recorded parameters."""
    from dvq_hip import synth
    from dvq_hip.pipeline import GraspGenerator
    from dvq_hip.tta import GraspRefiner
    dev = torch.device("cuda", 0)
    gen = GraspGenerator.synthetic(dev, seed=0, dtype=torch.float32)
    g = np.random.default_rng(2)
    pc = grasp_io.make_obj_pc(g.uniform(-0.05, 0.05, size=(3000, 3)), 0.17)
    plain, _ = grasp_io.diverse_grasps(gen, pc, n_grasps=4, rng=np.random.RandomState(5), first_id=40, device=dev)
    refiner = GraspRefiner(gen.rh_mano, faces=synth.closed_hand_faces(0), iters=5)
    rec, verts = grasp_io.diverse_grasps(gen, pc, n_grasps=4, rng=np.random.RandomState(5), first_id=40, device=dev, refiner=refiner)
    p0 = torch.tensor(plain["recon_params"], dtype=torch.float32).reshape(4, 61)
    p1 = torch.tensor(rec["recon_params"], dtype=torch.float32).reshape(4, 61)
    pl = grasp_io.random_placements(4, np.random.RandomState(5))
    clouds = grasp_io.place_clouds(pc, pl["R"]).to(dev)
    want = refiner.refine(clouds, p0.to(dev))["recon_param"]
    assert torch.allclose(p1, want.cpu(), rtol=0, atol=1e-6)
    assert torch.equal(p1[:, :10], p0[:, :10])                              # shape coefficients are not adapted
    hand = gen.rh_mano(global_orient=want[:, 10:13], hand_pose=want[:, 13:58], transl=want[:, 58:]).vertices
    assert torch.allclose(verts, hand, rtol=0, atol=1e-6)


def test_reader_and_placements_against_the_reference_stored_outputs():
    """tests/golden/hmdo_obj_id_0.json is one file the reference's generation script wrote
    (DVQ-VAE-2/diverse_grasp/HMDO/obj_id_0.json, gen_diverse_grasp_ho3d.py:322-331); mano_hmdo.npz holds the
    placements of all twelve.  The reader must return its arrays, and every stored rotation must be the
    Rx Ry Rz product of its stored Euler angles (gen_diverse_grasp_ho3d.py:214-222)."""
    import os
    from conftest import GOLDEN, load_golden
    from dvq_hip import grasp_io
    g = load_golden("mano_hmdo")
    rec = grasp_io.read_grasp_json(os.path.join(GOLDEN, "hmdo_obj_id_0.json"))
    assert rec["recon_params"].shape == (10, 61) and rec["R"].shape == (10, 3, 4)
    assert rec["trans"].shape == (10, 3, 1) and rec["r"].shape == (10, 3)
    np.testing.assert_array_equal(rec["recon_params"], g["recon_params"][:10])
    np.testing.assert_array_equal(rec["R"], g["R"][:10])
    for i in range(120):
        np.testing.assert_allclose(grasp_io.euler_rotation(g["r"][i]), g["R"][i][:, :3], rtol=0, atol=1e-15)
        # (the translations are not the bare constant of gen_diverse_grasp_ho3d.py:221: the script that wrote these
        # files also subtracted an object mean, as gen_diverse_grasp_FHAB.py:210 does; only R = [rot | trans] holds)
        np.testing.assert_array_equal(g["trans"][i][:, 0], g["R"][i][:, 3])
    # a record written back reads the same
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "o.json")
        grasp_io.write_grasp_json(path, grasp_io.grasp_record(torch.from_numpy(rec["recon_params"]),
                                                              dict(R=rec["R"], trans=rec["trans"], r=rec["r"])))
        back = grasp_io.read_grasp_json(path)
    for k in rec:
        np.testing.assert_array_equal(back[k], rec[k])
