"""Pin oracle/gru4rec_oracle.py against golden vectors produced by running the reference's GRU4Rec (hidden 128 fixture)."""
import os

import numpy as np
import torch

from oracle import gru4rec_oracle as GO
from oracle import sasrec_oracle as O


def test_gru4rec_forward_grads_adam_eval(golden_dir):
    z = np.load(os.path.join(golden_dir, "gru4rec_d64.npz"))
    g = {k: z[k] for k in z.files}
    p = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    b = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    nl = int(g["meta.layer_num"])
    loss, q, grads = GO.grads_of(p, b, nl)
    np.testing.assert_allclose(q.numpy(), g["out.query"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(float(loss), float(g["out.loss"]), rtol=2e-6)
    for k, gv in grads.items():
        ref = g["grad." + k]
        assert float(np.abs(gv.numpy() - ref).max()) < 3e-4 * max(1e-8, float(np.abs(ref).max())), k
    params = {k: v.clone() for k, v in p.items() if k != "query_encoder.0.1.weight"}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    new = O.adam_step(params, grads, m, v, 1, lr=float(g["meta.lr"]), wd=float(g["meta.weight_decay"]))
    for k in new:
        well = np.abs(g["grad." + k]) > 1e-4
        np.testing.assert_allclose(new[k].numpy()[well], g["adam1." + k][well], rtol=0, atol=5e-6)
    ql = GO.gru4rec_encode(p, torch.from_numpy(g["eval.in_item_id"]), torch.from_numpy(g["eval.seqlen"]), nl, "last")
    np.testing.assert_allclose(ql.numpy(), g["eval.query_last"], rtol=2e-4, atol=2e-6)
