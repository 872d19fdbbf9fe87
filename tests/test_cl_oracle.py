"""Pin oracle/cl4srec_oracle.py against the golden vectors produced by RUNNING the reference's CL4SRec (views recorded).  CPU only."""
import os

import numpy as np
import torch

from oracle import cl4srec_oracle as CO


def load_cl(golden_dir, name="cl4srec_d64"):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    g = {k: z[k] for k in z.files}
    p = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    p.pop("query_encoder.item_encoder.weight", None)
    batch = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    views = ((torch.from_numpy(g["view.i"]), torch.from_numpy(g["view.i_len"])), (torch.from_numpy(g["view.j"]), torch.from_numpy(g["view.j_len"])))
    cfg = {"H": int(g["meta.head_num"]), "n_layer": int(g["meta.layer_num"]), "eps": float(g["meta.layer_norm_eps"]),
           "temperature": float(g["meta.temperature"]), "cl_weight": float(g["meta.cl_weight"])}
    return g, p, batch, views, cfg


def test_views_are_legal_augmentations(golden_dir):
    g, p, batch, views, cfg = load_cl(golden_dir)
    n_items = int(g["meta.num_items"])
    assert p["item_embedding.weight"].shape[0] == n_items + 1                 # the mask item
    for seq, ln in views:
        for b in range(seq.shape[0]):
            n0, n1 = int(batch["seqlen"][b]), int(ln[b])
            src = batch["in_item_id"][b, :n0].tolist()
            v = seq[b, :n1].tolist()
            assert (seq[b, n1:] == 0).all()
            if n1 != n0:                                                     # crop
                assert n1 == CO.crop_len(n0, float(g["meta.tau"])) and any(src[s:s + n1] == v for s in range(n0 - n1 + 1))
            elif n_items in v:                                               # mask
                assert sum(x == n_items for x in v) == CO.mask_count(n0, float(g["meta.gamma"]))
                assert all(x == y or x == n_items for x, y in zip(v, src))
            else:                                                            # reorder (or an identity crop/mask of a short sequence)
                assert sorted(v) == sorted(src)


def test_contrastive_branch_and_total_gradient(golden_dir):
    g, p, batch, views, cfg = load_cl(golden_dir)
    P = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, bce, cl, oi, oj = CO.training_loss(P, batch, views, cfg)
    np.testing.assert_allclose(oi.detach().numpy(), g["out.view_i_mean"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(oj.detach().numpy(), g["out.view_j_mean"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(float(cl), float(g["out.cl_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(bce), float(g["out.bce_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(loss), float(g["out.loss"]), rtol=1e-5)
    keep = batch["seqlen"] != 1
    rows = CO.infonce(oi[keep].detach(), oj[keep].detach(), cfg["temperature"], reduce=False)
    np.testing.assert_allclose(rows.numpy(), g["out.cl_loss_rows"], rtol=2e-5, atol=1e-7)
    loss.backward()
    for k, v in P.items():
        ref = g["grad." + k]
        gv = v.grad.numpy() if v.grad is not None else np.zeros(v.shape, np.float32)
        err = float(np.abs(gv - ref).max()) / max(1e-8, float(np.abs(ref).max()))
        assert err < 2e-4, (k, err)
