"""MANO on the constants of the official model file, where that file is present (the build container only).

The reference tree ships MANO_RIGHT.pkl and closed_mano_faces.pkl (DVQ-VAE/models/mano/).  They are under the MANO
licence, so nothing derived from them is committed: the expected vertices are produced on the spot by running the
reference's own MANO class in a child process (tests/golden/make_golden.py mano_real) into a temporary directory.
Both files are pickles; they are read as data by dvq_hip.mano_io, which neither imports nor calls anything a file
names.  Skipped wherever the files are absent (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from dvq_hip import mano_io, synth
from oracle import dvq_oracle as O

MANO_DIR = "/root/reference/DVQ-VAE/models/mano"
RIGHT = os.path.join(MANO_DIR, "MANO_RIGHT.pkl")
CLOSED = os.path.join(MANO_DIR, "closed_mano_faces.pkl")
pytestmark = pytest.mark.skipif(not (os.path.exists(RIGHT) and os.path.exists(CLOSED)),
                                reason="the reference's MANO model files are not on this machine")


@pytest.fixture(scope="module")
def real():
    return mano_io.read_model_file(RIGHT)


@pytest.fixture(scope="module")
def reference_output(tmp_path_factory):
    out = tmp_path_factory.mktemp("mano_real")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "mano_real", "--out", str(out)],
                   check=True, timeout=600)
    with np.load(os.path.join(out, "mano_real.npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def test_model_file_fields(real):
    shapes = {"v_template": (778, 3), "shapedirs": (778, 3, 10), "posedirs": (778, 3, 135), "J_regressor": (16, 778),
              "weights": (778, 16), "hands_components": (45, 45), "hands_mean": (45,), "kintree_table": (2, 16),
              "f": (1538, 3)}
    for k, shp in shapes.items():
        assert tuple(real[k].shape) == shp, k
    parents = mano_io.to_np(real["kintree_table"], np.int64)[0]
    assert list(parents[1:]) == list(synth.MANO_PARENTS[1:])          # the tree the kernels hard-code


def test_oracle_vs_reference_on_the_real_constants(real, reference_output):
    """oracle.mano_forward with the official constants vs the reference class (mano/model.py:311-359, lbs.py:33-120)
    on the 120 stored HMDO parameter rows + rest pose + two large-angle rows."""
    g = reference_output
    p = torch.from_numpy(g["recon_params"])
    assert p.shape == (123, 61)
    sd = synth.mano_state_dict(batch_size=1, data=real)
    v, j, _ = O.mano_forward(sd, p[:, 10:13], p[:, 13:58], p[:, 58:61])
    scale = np.abs(g["vertices"]).max()
    err_v = np.abs(v.numpy() - g["vertices"]).max() / scale
    err_j = np.abs(j.numpy() - g["joints"]).max() / scale
    print(f"real MANO constants: oracle vs reference, relative error vertices {err_v:.2e}, joints {err_j:.2e}")
    assert err_v <= 1e-5 and err_j <= 1e-5


def test_skinning_table_holds_every_real_weight(real):
    """The skinning kernel's table form has eight slots per vertex; the official weights need at most six, and the
    table rebuilds the dense matrix exactly."""
    from mano.model import skinning_table
    w = torch.from_numpy(mano_io.to_np(real["weights"]))
    nnz = (w != 0).sum(1)
    assert int(nnz.max()) <= 8 and int(nnz.min()) >= 1
    fits, joints, weights = skinning_table(w, 8)
    assert fits
    dense = torch.zeros_like(w)
    dense.scatter_add_(1, joints.long(), weights)
    assert torch.equal(dense, w)
    for v in range(w.shape[0]):                                       # joints ascending within the used slots
        js = joints[v, :int(nnz[v])].tolist()
        assert js == sorted(js)
    print(f"real skinning weights: {int(nnz.min())}-{int(nnz.max())} non-zeros per vertex, mean {float(nnz.float().mean()):.2f}")


def test_product_layer_reads_the_real_file(real):
    """mano.load on the official file (constructor only: no device needed) registers the reference's 14 tensors with
    the reference's values."""
    import mano
    m = mano.load(RIGHT, use_pca=True, num_pca_comps=45, batch_size=1, flat_hand_mean=True)
    assert m.is_rhand and m.faces.shape == (1538, 3)
    want = synth.mano_state_dict(batch_size=1, data=real)
    got = m.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(got[k], want[k]), k
    left = mano.load(os.path.join(MANO_DIR, "MANO_LEFT.pkl"), use_pca=True, num_pca_comps=45, flat_hand_mean=True)
    assert not left.is_rhand
    byfolder = mano.load(MANO_DIR, is_rhand=True, use_pca=True, num_pca_comps=45, flat_hand_mean=True)
    assert torch.equal(byfolder.v_template, m.v_template)


def test_closed_hand_faces(real):
    """closed_mano_faces.pkl (gen_HDMO_TTA.py:61-62): (1552, 3) triangles over the 778 vertices that contain the
    model's own 1538 faces' vertex set and close the wrist -- every edge is shared by exactly two faces."""
    from dvq_hip.tta import read_closed_faces
    assert np.array_equal(read_closed_faces(CLOSED), mano_io.read_pickled_arrays(CLOSED).astype(np.int64))
    f = mano_io.read_pickled_arrays(CLOSED)
    assert isinstance(f, np.ndarray) and f.shape == (1552, 3) and np.issubdtype(f.dtype, np.integer)
    f = f.astype(np.int64)
    assert f.min() >= 0 and f.max() < 778
    edges = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, counts = np.unique(edges, axis=0, return_counts=True)
    assert (counts == 2).all()
    open_f = mano_io.to_np(real["f"], np.int64)
    assert np.array_equal(f[:1538], open_f)                           # the model's faces, then 14 over the wrist loop
    assert len(set(f[1538:].ravel().tolist())) == 16
    e2 = np.sort(np.concatenate([open_f[:, [0, 1]], open_f[:, [1, 2]], open_f[:, [2, 0]]]), axis=1)
    _, c2 = np.unique(e2, axis=0, return_counts=True)
    assert (c2 == 1).sum() > 0                                        # the model's own mesh is open at the wrist
