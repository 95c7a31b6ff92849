"""Host-side logic that needs no GPU: sharding + all-gather over gloo (world_size 2),
counter-based sampling, state_dict layouts, the C-ABI exports and the MANO file reader."""
import ctypes
import os
import pickle
import re
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT
from dvq_hip import parallel, rng, synth


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 16, 4096, 4099):
        for ws in (1, 2, 3, 4, 8):
            spans = [parallel.shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) <= (n + ws - 1) // ws


def test_uniforms_do_not_depend_on_sharding():
    ids = torch.arange(1000)
    u = rng.uniforms(ids, 9, seed=3)
    assert u.shape == (1000, 9) and float(u.min()) >= 0 and float(u.max()) < 1
    assert torch.equal(rng.uniforms(ids[250:500], 9, seed=3), u[250:500])
    assert not torch.equal(rng.uniforms(ids, 9, seed=4), u)
    assert abs(float(u.mean()) - 0.5) < 0.02
    p = torch.tensor([[0.1, 0.2, 0.7], [0.0, 1.0, 0.0]], dtype=torch.float32)
    assert rng.sample_rows(p, torch.tensor([0.05, 0.5])).tolist() == [0, 1]
    assert rng.sample_rows(p, torch.tensor([0.95, 0.999])).tolist() == [2, 1]
    draws = rng.sample_rows(p[:1].expand(20000, 3), rng.uniforms(torch.arange(20000), 1, 0)[:, 0])
    freq = torch.bincount(draws, minlength=3).float() / 20000
    assert torch.allclose(freq, p[0], atol=0.02)


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = parallel.shard_range(n, rank, world)
    per = (n + world - 1) // world
    objs = synth.object_clouds(e - b, 64, seed=5, first=b)           # this rank's shard of the object list
    u = rng.uniforms(torch.arange(b, e), 9, seed=1)
    # stand-in for the per-object computation: anything that is a function of the object alone
    verts = torch.stack([objs.sum(dim=(1, 2)), u.sum(dim=1).float()], 1).unsqueeze(-1).expand(-1, -1, 3).contiguous()
    out = parallel.gather_rows(verts, n_total=n)                      # the short last shard is padded inside
    # the asynchronous form (the bench keeps one step's gather in flight beside the next step): two collectives
    # outstanding at once, complete after their wait()
    out_a, wa = parallel.gather_rows(parallel.pad_rows(verts, per), n_total=n, async_op=True)
    out_b, wb = parallel.gather_rows(parallel.pad_rows(verts * 2, per), n_total=n, async_op=True)
    wa.wait()
    wb.wait()
    assert torch.equal(out_a, out) and torch.equal(out_b, out * 2)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_process():
    n, world = 11, 2                                                  # odd: the last shard is short
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    objs = synth.object_clouds(n, 64, seed=5)
    u = rng.uniforms(torch.arange(n), 9, seed=1)
    want = torch.stack([objs.sum(dim=(1, 2)), u.sum(dim=1).float()], 1).unsqueeze(-1).expand(-1, -1, 3)
    assert got.shape == (n, 2, 3)
    np.testing.assert_array_equal(got, want.numpy())


class _StubModel:
    """Stands in for GenNet on the CPU: any function of the object and its uniforms alone."""
    rh_mano = object()
    GatedPixelCNN = None

    def gen(self, obj, x_tilde=None, uniforms=None):
        s = obj.sum(dim=(1, 2)).float().unsqueeze(1)
        return s + uniforms.float().sum(1, keepdim=True) + torch.arange(55.0), s * 2 + torch.arange(6.0)


class _StubHand:
    """A hand layer with another vertex count than MANO's 778: an empty shard must still gather blocks of this shape."""
    V = 50
    v_template = torch.zeros(V, 3)

    def __call__(self, betas, global_orient, hand_pose, transl):
        v = hand_pose.sum(1).reshape(-1, 1, 1) + torch.arange(self.V * 3.0).reshape(1, self.V, 3) + transl.unsqueeze(1)
        return types.SimpleNamespace(vertices=v)


def _generator_worker(rank, world, port, n, q):
    from dvq_hip.pipeline import GraspGenerator
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = parallel.shard_range(n, rank, world)
    objs = synth.object_clouds(n, 64, seed=5)[b:e]
    g = GraspGenerator(_StubModel(), _StubHand(), seed=1)
    out = g.generate(objs, first_id=b, gather=True, n_total=n)
    out2 = g.generate(objs, first_id=b, gather=True, n_total=n, async_gather=True)
    out3 = g.generate(objs, first_id=b, gather=True, n_total=n, async_gather=True)     # completes out2's gathers
    assert torch.equal(out2["vertices"], out["vertices"]) and torch.equal(out2["recon_param"], out["recon_param"])
    g.finish()
    assert torch.equal(out3["vertices"], out["vertices"]) and not g._in_flight
    if rank == world - 1:
        with pytest.raises(RuntimeError, match="its shard"):
            parallel.gather_rows(torch.zeros(e - b + 7, 3), n_total=n)
    if rank == 0:
        q.put((out["vertices"].numpy(), out["recon_param"].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_and_empty_shards_through_the_generator():
    """10 objects over 4 ranks: shards of 3, 3, 3, 1 -- and 9 over 4: the last rank holds nothing -- go through
    GraspGenerator.generate(gather=True) and come back as the one-process result, padding trimmed."""
    from dvq_hip.pipeline import GraspGenerator
    for n, world in ((10, 4), (9, 4)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 31500 + (os.getpid() + n) % 2000
        procs = [ctx.Process(target=_generator_worker, args=(r, world, port, n, q)) for r in range(world)]
        for p in procs:
            p.start()
        verts, param = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        want = GraspGenerator(_StubModel(), _StubHand(), seed=1).generate(synth.object_clouds(n, 64, seed=5))
        assert verts.shape == (n, _StubHand.V, 3) and param.shape == (n, 61)
        np.testing.assert_array_equal(verts, want["vertices"].numpy())
        np.testing.assert_array_equal(param, want["recon_param"].numpy())


def test_state_dict_layouts_match_the_reference_counts():
    from network.DVQVAE import DVQVAE
    from network.gen_net import GenNet
    import mano
    from dvq_hip.mano_io import Struct
    g = GenNet()
    keys = list(g.state_dict().keys())
    assert len(keys) == 333 and keys == list(synth.gennet_state_dict(0).keys())
    assert sum(p.numel() for p in g.parameters()) == 100_495_576     # 100.50 M, as the instantiated reference
    assert sum(1 for k in keys if k.startswith("GatedPixelCNN.")) == 140
    assert sum(1 for k in keys if k.startswith("obj_encoder_type.")) == 58
    d = DVQVAE(obj_inchannel=4)
    assert list(d.state_dict().keys()) == list(synth.dvqvae_state_dict(0).keys())
    rh = mano.load('', data_struct=Struct(**synth.mano_model_data(0)), use_pca=True, num_pca_comps=45, batch_size=1,
                   flat_hand_mean=True)
    g.set_rh_mano(rh)
    extra = [k for k in g.state_dict().keys() if k.startswith("rh_mano.")]
    assert extra == ["rh_mano." + k for k in ("betas", "global_orient", "transl", "hand_pose", "faces_tensor", "v_template",
                                                "shapedirs", "J_regressor", "posedirs", "parents", "lbs_weights",
                                                "hand_components", "hand_mean", "pose_mean")]
    assert rh.faces.shape == (1538, 3)
    # the scripts' loading convention (gen_diverse_grasp_obman.py:333-338)
    ckpt = {"network": synth.gennet_state_dict(1), "epoch": 3}
    model_dict = g.state_dict()
    model_dict.update({k: v for k, v in ckpt["network"].items() if k in model_dict})
    g.load_state_dict(model_dict)


def test_library_exports_every_declared_symbol():
    from dvq_hip import _lib
    header = open(os.path.join(ROOT, "include", "dvq_hip.h")).read()
    declared = set(re.findall(r"\b(dvq_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert os.path.exists(_lib.LIB_PATH), "build the library first: make -C d-vqvae_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.dvq_version.restype = ctypes.c_int
    assert lib.dvq_version() == 113


def test_cpu_tensors_are_refused_everywhere():
    from dvq_hip import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.vq_argmin(torch.zeros(4, 256), torch.zeros(128, 256))
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 16), torch.zeros(8, 16), None)


def test_mano_file_reader_builds_arrays_and_refuses_code(tmp_path):
    """A synthetic pickle with the class layout of the official MANO files: the reader takes the arrays out of the
    chumpy and scipy records without importing or calling anything the file names, and refuses every other name."""
    from dvq_hip import mano_io
    ch = types.ModuleType("chumpy"); chch = types.ModuleType("chumpy.ch"); chre = types.ModuleType("chumpy.reordering")

    class Ch:
        def __init__(self, x=None): self.x = x
        def __getstate__(self): return {"x": self.x}
        def __setstate__(self, s): self.__dict__.update(s)

    class Select:
        def __init__(self, a=None, idxs=None, shape=None): self.a, self.idxs, self.preferred_shape = a, idxs, shape
        def __getstate__(self): return {"a": self.a, "idxs": self.idxs, "preferred_shape": self.preferred_shape}
        def __setstate__(self, s): self.__dict__.update(s)

    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    Select.__module__, Select.__qualname__ = "chumpy.reordering", "Select"
    chch.Ch, chre.Select = Ch, Select
    saved = {k: sys.modules.get(k) for k in ("chumpy", "chumpy.ch", "chumpy.reordering")}
    sys.modules.update({"chumpy": ch, "chumpy.ch": chch, "chumpy.reordering": chre})
    try:
        d = synth.mano_model_data(0)
        full = np.arange(778 * 3 * 12, dtype=np.float64).reshape(778, 3, 12)
        blob = dict(d)
        blob_plain = dict(d)
        blob["v_template"] = Ch(d["v_template"])
        blob["shapedirs"] = Select(Ch(full), np.arange(778 * 3 * 12).reshape(778, 3, 12)[:, :, :10].ravel(), (778, 3, 10))
        path = tmp_path / "MANO_RIGHT.pkl"
        path.write_bytes(pickle.dumps(blob, protocol=2))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    out = mano_io.read_model_file(str(path))
    np.testing.assert_array_equal(mano_io.to_np(out["v_template"]), d["v_template"].astype(np.float32))
    np.testing.assert_array_equal(mano_io.to_np(out["shapedirs"]), full[:, :, :10].astype(np.float32))
    np.testing.assert_array_equal(mano_io.to_np(out["weights"]), d["weights"].astype(np.float32))
    evil = tmp_path / "MANO_LEFT.pkl"
    evil.write_bytes(pickle.dumps({"v_template": os.getcwd}, protocol=2))
    with pytest.raises(mano_io.ModelFileError):
        mano_io.read_model_file(str(evil))

    class Runs:                                    # a file that asks for a call when it is loaded
        def __reduce__(self):
            return (os.mkdir, (str(tmp_path / "made_by_the_file"),))
    for proto in (2, 4):
        evil.write_bytes(pickle.dumps(dict(blob_plain, v_template=Runs()), protocol=proto))
        with pytest.raises(mano_io.ModelFileError):
            mano_io.read_model_file(str(evil))
        assert not (tmp_path / "made_by_the_file").exists()      # recorded as a tag, never called
    # every protocol Python 3 writes holds the same arrays
    for proto in (2, 3, 4):
        path_p = tmp_path / f"plain{proto}.pkl"
        path_p.write_bytes(pickle.dumps(blob_plain, protocol=proto))
        out_p = mano_io.read_model_file(str(path_p))
        for k in mano_io.FIELDS:
            np.testing.assert_array_equal(mano_io.to_np(out_p[k], np.float64), mano_io.to_np(d[k], np.float64))
    import mano
    m = mano.load(str(path), use_pca=True, num_pca_comps=45, batch_size=1, flat_hand_mean=True)
    assert m.is_rhand and m.v_template.shape == (778, 3)
    # damaged files: cut short or with flipped bytes, every outcome is either the arrays or ModelFileError -- no other
    # exception, and nothing the bytes happen to name is ever looked up
    small = pickle.dumps({k: (np.asarray(v)[:4] if isinstance(v, np.ndarray) else v) for k, v in blob_plain.items()}, protocol=2)
    r = np.random.default_rng(0)
    bad = tmp_path / "damaged.pkl"
    for trial in range(300):
        data = bytearray(small)
        if trial % 3 == 0:
            data = data[:int(r.integers(1, len(data)))]
        else:
            for _ in range(int(r.integers(1, 6))):
                data[int(r.integers(0, len(data)))] = int(r.integers(0, 256))
        bad.write_bytes(bytes(data))
        try:
            mano_io.read_pickled_arrays(str(bad))
        except mano_io.ModelFileError:
            pass


def test_committed_bench_line_follows_the_contract():
    """The newest bench line kept under profiles/ carries every field the measurement contract names."""
    import glob
    import json
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert paths, "no bench line under profiles/"
    line = open(paths[-1]).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "grasps generated/sec" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - d["n_gpus"] * d["config"]["objects_per_gpu"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-3 * d["value"]
