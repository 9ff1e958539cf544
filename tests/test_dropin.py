"""Module-level drop-in (`install_as_mrgcn`) and the boundary padding of variable-length encodings."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from tests.util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mrgcn")), reason="reference tree not on this machine")
@pytest.mark.parametrize("order", ["first", "after_pkg", "after_leaf", "patched_first", "patched_after"])
def test_install_as_mrgcn_leaves_the_rest_of_the_reference_importable(order):
    """run.py:12-19 / tasks/node_classification.py:9-16: whatever is imported first, the task modules
    come from the reference, the seven replaced leaves from this package, and the reference's own
    build_model builds this package's MRGCN."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dropin_child.py"), order],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"dropin ok {order}" in r.stdout


def test_install_as_mrgcn_without_a_reference_installation():
    """No reference on sys.path: the leaves still import under the reference's names."""
    code = ("import sys; sys.path.insert(0, %r); import mrgcn_amd; mrgcn_amd.install_as_mrgcn();"
            "from mrgcn.models.mrgcn import MRGCN; from mrgcn.data.batch import FullBatch;"
            "import mrgcn.layers.graph as g; assert g.__name__ == 'mrgcn_amd.layers.graph';"
            "import importlib.util as u; assert u.find_spec('mrgcn.tasks') is None; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd="/tmp")
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def _pad_inputs(g):
    tok_len = g["tok_len"]
    toks = np.empty(len(tok_len), dtype=object)
    off = 0
    for i, L in enumerate(tok_len):
        toks[i] = g["tok_flat"][off:off + L].copy()
        off += L
    mats = np.empty(len(g["mat_w"]), dtype=object)
    for i in range(len(mats)):
        mats[i] = sp.csr_matrix((g[f"mat{i}.data"], g[f"mat{i}.indices"], g[f"mat{i}.indptr"]),
                                shape=tuple(g[f"mat{i}.shape"]))
    return toks, tok_len, mats, g["mat_w"], g["num"]


@pytest.mark.parametrize("case,pads,width", [("default", {}, None), ("pad101", {"xsd.string": 101}, None),
                                             ("wide", {"xsd.string": 7}, 16)])
def test_pad_and_to_dense_match_the_reference(case, pads, width):
    """mrgcn/data/batch.py:25-68 (golden: tests/golden/make_pad_goldens.py)."""
    from mrgcn_amd.data.batch import FullBatch
    g = np.load(os.path.join(GOLDEN, "pad_batch.npz"))
    toks, tok_len, mats, mat_w, num = _pad_inputs(g)
    if width is not None:
        tok_len, mat_w = np.full(len(toks), width), np.full(len(mats), width)
    X = [np.zeros((20, 0), dtype=np.float32),
         ["xsd.string", [[toks, np.arange(len(toks)), tok_len]], False],
         ["ogc.wktLiteral", [[mats, np.arange(len(mats)), mat_w]], False],
         ["xsd.numeric", [[num, np.arange(len(num)), np.full(len(num), 4)]], False]]
    b = FullBatch(None, X, np.arange(20))
    b.pad_(pad_symbols=pads)
    tp = b.X[1][1][0][0]
    assert tp.dtype == g[f"{case}.tok_padded"].dtype and np.array_equal(tp, g[f"{case}.tok_padded"])
    assert np.array_equal(np.array([m.shape for m in b.X[2][1][0][0]]), g[f"{case}.mat_shapes"])
    assert b.X[3][1][0][0] is num  # fixed-width encodings are left alone
    b.to_dense_()
    md = b.X[2][1][0][0]
    assert md.dtype == g[f"{case}.mat_dense"].dtype and np.array_equal(md, g[f"{case}.mat_dense"])
    b.A = sp.identity(20, format="csr", dtype=np.float32)
    b.as_tensors_()  # padded / densified members are plain arrays: torch.from_numpy takes them
    assert tuple(b.X[1][1][0][0].shape) == tp.shape and tuple(b.X[2][1][0][0].shape) == md.shape


def test_pad_rejects_a_sequence_longer_than_the_width():
    from mrgcn_amd.data.batch import pad_token_sequences
    seqs = np.empty(2, dtype=object)
    seqs[0], seqs[1] = np.arange(1200), np.arange(3)
    with pytest.raises(ValueError):
        pad_token_sequences(seqs, 0, 5)
