"""Pin oracle/fmlp_oracle.py (circular-convolution restatement of the spectral filter) against golden vectors from the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import fmlp_oracle as FO
from oracle import sasrec_oracle as O


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "fmlp_d64.npz"))
    g = {k: z[k] for k in z.files}
    params = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    batch = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    return g, params, batch


@pytest.mark.parametrize("use_fft", [False, True])
def test_fmlp_forward(golden_dir, use_fft):
    g, p, b = load(golden_dir)
    nl = int(g["meta.layer_num"])
    q, acts = FO.fmlp_encode(p, b["in_item_id"], nl, use_fft=use_fft, return_all=True)
    for k in ("x0", "filter0", "layer0", "filter1", "layer1"):
        np.testing.assert_allclose(acts[k].numpy(), g["act." + k], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(q.numpy(), g["out.query"], rtol=2e-4, atol=2e-5)
    loss, pos, ng = O.score_bce(q, p["item_embedding.weight"], b["item_id"], b["neg_item"], True)
    np.testing.assert_allclose(float(loss), float(g["out.loss"]), rtol=2e-6)
    lnr, _, _ = O.score_bce(q, p["item_embedding.weight"], b["item_id"], b["neg_item"], False)
    np.testing.assert_allclose(lnr.numpy(), g["out.loss_noreduce"], rtol=2e-4, atol=1e-7)


def test_fmlp_gradients_and_adam(golden_dir):
    g, p, b = load(golden_dir)
    nl = int(g["meta.layer_num"])
    loss, q, grads = FO.grads_of(p, b, nl)
    for k, gv in grads.items():
        ref = g["grad." + k]
        assert float(np.abs(gv.numpy() - ref).max()) < 3e-4 * max(1e-8, float(np.abs(ref).max())), k
    # imaginary parts of the DC and Nyquist bins are ignored by irfft: zero gradient in the reference too
    gc = g["grad.item_encoder.layer.0.filterlayer.complex_weight"]
    assert np.abs(gc[0, 0, :, 1]).max() == 0 and np.abs(gc[0, 25, :, 1]).max() < 1e-9
    params = {k: v.clone() for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    for t, tag in ((1, "adam1."), (2, "adam2.")):
        _, _, gr = FO.grads_of(params, b, nl)
        params = O.adam_step(params, gr, m, v, t, lr=float(g["meta.lr"]))
        for k in params:
            well = np.abs(g["grad." + k]) > 1e-5
            np.testing.assert_allclose(params[k].numpy()[well], g[tag + k][well], rtol=0, atol=5e-6)
            np.testing.assert_allclose(params[k].numpy(), g[tag + k], rtol=0, atol=2.1e-3)   # <= 2 Adam steps of lr: ill-conditioned |g|~eps elements
