#!/usr/bin/env python3
"""bench.py — training sequences/sec of the SASRec hot path (BASELINE.json metric) on N MI355X.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N ...                          (stand-alone: re-executes itself under torch.distributed.run, N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the reference's training iteration (model/basemodel.py:193-199) over one batch of
B = 256 rows per GPU (configs/basemodel.yaml:2) of synthetic Amazon-toys-shaped data (N = 11 925 items,
L = 50, d = 64, toys seqlen histogram): device-side batch selection, in-kernel negative sampling, SASRec
forward with dropout 0.5, tied scorer + BCE, full backward, [sum-all-reduce of the flat gradient over RCCL
when N > 1], dense Adam.  Inputs (dataset tensors + permutation) are resident in HBM before timing starts.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_match  # noqa: E402  (tools/pmc_match.py: launch kind -> the one rocprofv3 kernel name of the regime)

PROFILE_ROUND = 6              # profiles/round<N>_* files this bench refers to (tools/refresh_profiles.sh)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def init_params_like_reference(eng, seed):
    """utils/utils.py:70-81 normal_initialization (N(0,0.02) embeddings/linears, zero biases, LN (1,0), PAD row 0);
    in_proj_weight keeps torch's xavier_uniform (it is a bare Parameter) and is identical in all layers."""
    g = torch.Generator().manual_seed(seed)
    D = eng.D
    bound = (6.0 / (3 * D + D)) ** 0.5
    in_proj = (torch.rand(3 * D, D, generator=g) * 2 - 1) * bound
    sd = {}
    for name, shp in zip(eng.names, eng.shapes):
        if name.endswith("in_proj_weight"):
            sd[name] = in_proj.clone()
        elif name.endswith(("norm1.weight", "norm2.weight")):
            sd[name] = torch.ones(shp)
        elif name.endswith("bias"):
            sd[name] = torch.zeros(shp)
        else:
            sd[name] = 0.02 * torch.randn(shp, generator=g)
    sd["item_embedding.weight"][0] = 0
    eng.load_named(sd)


def kernel_flops(kind, T, B, L, D, F, n_layer, seqlen, big=None, in_tile=False):
    """ALGORITHMIC flops of one launch on the packed batch: 2*M*N*K per GEMM over the T valid tokens.
    in_tile: the attention of the layer runs inside the token-tile launches (csrc/attn_tile.h): forward 2, backward 5 products per pair."""
    att = 2.0 * float((seqlen * (seqlen + 1) // 2).sum()) * D if in_tile else 0.0
    if kind in ("post_fwd", "post_bwd"):
        return 2.0 * T * (D * D + 2 * D * F) + (2.0 * T * 3 * D * D if n_layer > 1 else 0.0) + (2.0 if kind == "post_fwd" else 5.0) * att   # + the next layer's qkv projection
    if kind == "post_mid":
        return 2.0 * 2.0 * T * (D * D + 2 * D * F) + 7.0 * att          # post_fwd + post_bwd of the last layer (scorer: VALU dots, not counted)
    if kind in ("qkv_fwd", "qkv_bwd", "embqkv_fwd", "qkv_embed_bwd"):
        return 2.0 * T * 3 * D * D
    if kind in ("wgrad", "wgrad_fused"):
        return 2.0 * T * (4 * D * D + 2 * D * F) * n_layer + (2.0 * T * 3 * D * D if kind == "wgrad_fused" and not (B * L > 16384 if big is None else big) else 0.0)
    if kind in ("attn_fwd", "attn_bwd"):
        pairs = float((seqlen * (seqlen + 1) // 2).sum())
        return (2.0 if kind == "attn_fwd" else 5.0) * 2.0 * pairs * D      # QK^T + PV (fwd); +dP, dQ, dK, dV (bwd)
    return 0.0


def sasrec_kernel_rooflines(lib, _lib, plan, mw, out, args, B, L, D, F, NL, T_last, seqlen_last, group, step_s, dev):
    """per-kernel launch durations (HIP events on the launch stream, on the state the last step left in the workspace) of the fused
    SASRec step -> out["roofline"] (dominant MFMA kernel), out["roofline_step"], out["kernel_us_per_step"].  mw: ctypes
    dr4sr_meta_weighting for the MetaModel's weighted step (model/metamodel.py:174-194), else None."""
    # whole-step MFMA roofline: algorithmic flops of ONE rank's step on the tokens it really holds (linear layers fwd + data
    # grads + weight grads = 3 x, causal attention n(n+1)/2 pairs x {QK^T, PV} x 3.5 for fwd + bwd) over the measured step time
    lin = 3.0 * NL * (2 * D * 3 * D + 2 * D * D + 4 * D * F) * float(seqlen_last.sum())
    att = 3.5 * NL * 2 * 2 * D * float((seqlen_last.astype(np.float64) * (seqlen_last + 1) / 2).sum())
    out["roofline_step"] = {"bound": "mfma", "achieved": (lin + att) / step_s / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                            "frac": (lin + att) / step_s / 1e12 / MFMA_F32_PEAK_TF, "flops_per_step": lin + att,
                            "note": "all kernels of the step, last batch's token count"}
    # the launches dr4sr_sasrec_train_steps really enqueues per step: (kind, layer argument, launches / step).
    # post_fwd / post_bwd at layer 0 are the fused forms (they carry layer 1's qkv projection / its backward); the last
    # layer's post_fwd + scorer + post_bwd is ONE launch (post_mid).  In the latency regime (expected tokens of the plan
    # <= ~10 k: dr4sr_sasrec_at_scale) the embedding-stage backward rides in the k_wgrad launch (`wgrad_fused`), at scale
    # it is a launch of its own.
    big = bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & 1)
    in_tile = bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & 4)       # latency regime: no attention launches (csrc/attn_tile.h)
    fold_fwd = bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & 32)     # round 6: the attention forward runs inside the wave-tile forward launches
    launches = [("prep", 0, 1.0 / max(1, group)), ("embqkv_fwd", 0, 1)] + ([] if (in_tile or fold_fwd) else [("attn_fwd", NL - 1, NL)]) + [("post_fwd", 0, NL - 1),
                ("post_mid", 0, 1)] + ([] if in_tile else [("attn_bwd", NL - 1, NL)]) + [("post_bwd", 0, NL - 1)]
    launches += ([("qkv_embed_bwd", 0, 1)] if big else []) + [("wgrad_fused", 0, 1), ("adam", 0, 1)]
    per_step_launches = {k: n for k, _, n in launches}
    mwp = C.byref(mw) if mw is not None else None

    def launch(kind, layer):
        _lib.check(lib.dr4sr_sasrec_launch_kernel_weighted(C.byref(plan), mwp, _lib.KERNEL_IDS[kind], layer, _lib.cur_stream()), kind)
    ktime = {}
    reps = 50
    for kind, layer, _ in launches:
        for _ in range(5):
            launch(kind, layer)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            launch(kind, layer)
        b.record()
        b.synchronize()
        ktime[kind] = a.elapsed_time(b) * 1e3 / reps          # us per launch (back-to-back launches)
    step_us = {k: v * per_step_launches.get(k, 1) for k, v in ktime.items()}
    # the dominant KERNEL = the kernel name with the largest share of the step, as a rocprofv3 --stats table ranks them.  An attention
    # "kind" is one kernel per layer only while one workgroup per sequence runs; with the length-class lists (plan bit 1) a call is
    # three different kernels backward (1..8-token VALU class, 16-row, 64-row) and two forward: their summed time is no kernel's share
    # (round 4: the sum, 2 x 47 us at toys B = 8 192, tied with k_wt_post_mid's 95 us and made a three-kernel "kernel" the roofline's subject)
    # (round 4, plan bit 3: short-sequence plans at d = 64 run ONE window-attention launch per layer and direction instead — csrc/attn_tile_sa.hip)
    # (round 6, plan bit 4: where the lists would run, ONE wave-per-tile launch per layer and direction — csrc/attn_wave.hip)
    lists = bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & 2) and not bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & (8 | 16))
    names_per_kind = {"attn_bwd": 3 if lists else 1, "attn_fwd": 2 if lists else 1}
    dom = max((k for k in step_us if kernel_flops(k, 1, B, L, D, F, NL, seqlen_last, big) > 0),
              key=lambda k: step_us[k] / names_per_kind.get(k, 1))
    fl = kernel_flops(dom, T_last, B, L, D, F, NL, seqlen_last, big, in_tile)
    ach = fl / (ktime[dom] * 1e-6) / 1e12
    # HBM bytes per launch of that kernel from the PMC passes kept under profiles/ (tools/traffic_pmc.sh: FETCH_SIZE x2
    # gfx950 correction + WRITE_SIZE, separate rocprofv3 runs); only for the workloads that were profiled
    traffic = None
    tag = {(256, False): "B256_toys", (8192, False): "B8192_toys", (8192, True): "B8192_dense"}.get((B, bool(args.dense)))
    if mw is not None:
        tag = None                                         # the PMC passes were taken on the plain step
    traffic_src = None
    for rnd in range(PROFILE_ROUND, 0, -1):
        pj = os.path.join(ROOT, "profiles", "round%d_pmc_traffic_%s.json" % (rnd, tag)) if tag and D == 64 else None
        if pj and os.path.exists(pj):
            pm = json.load(open(pj))
            # exactly ONE kernel name per (launch kind, regime) — tools/pmc_match.py; round 5 summed k_post_mid + k_wt_post_mid here
            if dom.startswith("attn"):
                got = pmc_match.match_attention(pm, dom == "attn_bwd")
            else:
                got = pmc_match.match(pm, dom, big, D)[0]
            if got is not None:
                traffic, traffic_src = got, os.path.relpath(pj, ROOT)
                break
    # `traffic` is NOT measured by this run: rocprofv3 --pmc passes cannot run inside the bench; it is the per-launch HBM
    # byte count of the same kernel on the same workload from the committed PMC profile named in `traffic_source`
    out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                       "frac": ach / MFMA_F32_PEAK_TF, "traffic": traffic, "traffic_source": traffic_src,
                       "us_per_launch": ktime[dom], "flops_per_launch": fl}
    # the ragged layout leaves the attention kernels far below the MFMA ridge (157.3 TF / 8 TB/s = 19.7 flop/B): also report
    # the kernel against the sloped part of the roofline, min(MFMA peak, intensity x HBM peak), from its algorithmic bytes
    # (forward: q, k, v in + ctx out; backward: q, k, v, dctx in + dq, dk, dv out; 4*D bytes per token each)
    arrays = {"attn_fwd": 4, "attn_bwd": 7}.get(dom)
    if arrays:
        ab = float(arrays * T_last * D * 4)
        ceil_tf = min(MFMA_F32_PEAK_TF, fl / ab * HBM_PEAK_GBS / 1e3)
        out["roofline"].update({"algorithmic_bytes_per_launch": ab, "flop_per_byte": fl / ab, "ceiling_at_intensity": ceil_tf,
                                "frac_of_ceiling": ach / ceil_tf,
                                "hbm_rate": ab / (ktime[dom] * 1e-6) / 1e9, "hbm_frac": ab / (ktime[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS})
    out["kernel_us_per_step"] = {k: round(v, 2) for k, v in step_us.items()}
    # the token-tile kernels against BOTH roofs: at scale they are bound by the saved-activation stream (HBM), not by the matrix pipe
    both = {}
    wave_attn = bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & 16)
    for k, us in ktime.items():
        kb = kernel_bytes(k, T_last, D, F, NL, in_tile)
        if kb is None and wave_attn and k in ("attn_fwd", "attn_bwd"):
            # the wave-per-tile launches (csrc/attn_wave.hip), fp32 words per token: forward q | k | v in, ctx + {row max, 1 / row sum} x 2 heads out;
            # backward q | k | v, dctx, statistics, row terms, keep bits in, dq | dk | dv out
            kb = 4.0 * T_last * ((3 * D + D + 4) if k == "attn_fwd" else (3 * D + D + 4 + 2 + 4 + 3 * D))
        if kb is None:
            continue
        fk = kernel_flops(k, T_last, B, L, D, F, NL, seqlen_last, big, in_tile)
        both[k] = {"us_per_launch": round(us, 2), "hbm_frac": round(kb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   "mfma_f32_frac": round(fk / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4), "algorithmic_bytes": kb}
    out["roofline_tile_kernels"] = both
    if dom in both and both[dom]["hbm_frac"] > out["roofline"]["frac"]:
        r = out["roofline"]
        r.update({"bound": "hbm", "achieved": both[dom]["algorithmic_bytes"] / (ktime[dom] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": both[dom]["hbm_frac"], "mfma_f32_frac": both[dom]["mfma_f32_frac"], "algorithmic_bytes_per_launch": both[dom]["algorithmic_bytes"]})
    return ktime


def gru_kernel_rooflines(lib, _lib, plan, out, T_last, D, Hh, NLg, step_s):
    """GRU4Rec (model/gru4rec.py:12-34, module/layers.py:117-136): launch durations of the recurrence kernels (the dominant launches of
    the step: the cooperative multi-CU recurrence at B <= 1536, the single-workgroup one above) by HIP events on the launch stream ->
    out["roofline"] on the recurrent GEMM's algorithmic flops over the valid tokens (gh = h W_hh^T: 2*3H*H per token forward,
    dh = dgh W_hh the same backward), out["roofline_step"] on SURVEY §8(d)'s 3 x (gi + gh + out projection) per valid token."""
    def launch(kid, layer):
        _lib.check(lib.dr4sr_gru4rec_launch_kernel(C.byref(plan), kid, layer, _lib.cur_stream()), "gru4rec_launch_kernel")
    ktime = {}
    reps = 20
    for name, kid in (("rec_fwd", 0), ("rec_bwd", 1), ("gemm_in", 2)):
        tot = 0.0
        for layer in range(NLg):
            for _ in range(3):
                launch(kid, layer)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                launch(kid, layer)
            b.record()
            b.synchronize()
            tot += a.elapsed_time(b) * 1e3 / reps
        ktime[name] = tot / NLg                              # us per launch, mean over the layers
    # two-layer plans: both forward recurrences (and layer 2's input projection) are ONE launch, the layer wavefront of csrc/gru_coop.hip
    wave = bool(lib.dr4sr_gru4rec_uses_wavefront(int(plan.B), Hh, NLg, int(plan.L)))
    if wave:
        for _ in range(3):
            launch(3, 0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            launch(3, 0)
        b.record()
        b.synchronize()
        ktime["rec_fwd_one_launch_per_layer"] = ktime["rec_fwd"]
        ktime["rec_fwd"] = a.elapsed_time(b) * 1e3 / reps / NLg          # per layer, as the other entries
    dom = "rec_bwd" if ktime["rec_bwd"] >= ktime["rec_fwd"] else "rec_fwd"
    fl = 2.0 * T_last * 3 * Hh * Hh
    ach = fl / (ktime[dom] * 1e-6) / 1e12
    coop = bool(lib.dr4sr_gru4rec_uses_cooperative(int(plan.B), Hh))
    if wave and dom == "rec_fwd":                           # the one-launch forward: 2 recurrent GEMMs + layer 2's input GEMM, whole launch
        fl, ktime_dom, nlaunch, kname = 3.0 * fl, ktime[dom] * NLg, 1, "k_gru_fwd_wave"
        ach = fl / (ktime_dom * 1e-6) / 1e12
    else:
        ktime_dom, nlaunch, kname = ktime[dom], NLg, ("k_gru_%s_coop" if coop else "k_gru_%s") % ("bwd" if dom == "rec_bwd" else "fwd")
        if coop and dom == "rec_bwd" and Hh == 256 and int(plan.B) <= 1536 and not os.environ.get("DR4SR_GRU_BWD_F32") \
                and bool(lib.dr4sr_gru4rec_uses_cooperative(min(int(plan.B), 256), Hh)) and min(int(plan.B), 256) <= 256:
            kname = "k_gru_bwd_coop_bf"                   # 16 slices per group: the bf16x3 BPTT with W_hh in registers (csrc/gru_coop.hip)
    traffic, traffic_src = pmc_traffic("gru4rec_B256", kname) if int(plan.B) == 256 else (None, None)
    out["roofline"] = {"kernel": kname, "bound": "mfma",
                       "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": traffic,
                       "traffic_source": traffic_src, "us_per_launch": ktime_dom, "flops_per_launch": fl, "launches_per_step": nlaunch,
                       "note": "recurrent GEMM of one layer on the valid tokens; the launch is a chain of max(seqlen) dependent "
                               "time steps (latency-bound, SURVEY §8d), not an MFMA-throughput kernel"}
    per_tok = 3.0 * (2 * 3 * Hh * D + (NLg - 1) * 2 * 3 * Hh * Hh + NLg * 2 * 3 * Hh * Hh + 2 * Hh * D)
    out["roofline_step"] = {"bound": "mfma", "achieved": per_tok * T_last / step_s / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                            "frac": per_tok * T_last / step_s / 1e12 / MFMA_F32_PEAK_TF, "flops_per_step": per_tok * T_last,
                            "note": "3 x (input GEMMs + recurrent GEMMs + output projection) on the last batch's valid tokens"}
    out["kernel_us_per_step"] = {k: round(v * NLg, 2) for k, v in ktime.items()}
    if wave:
        out["kernel_us_per_step"]["note"] = ("rec_fwd = k_gru_fwd_wave: both layers' forward recurrences + gi_2 in one launch "
                                             "(gemm_in then only runs for layer 1: half the figure above)")


def pmc_traffic(tag, kernel_prefix):
    """(HBM bytes per launch, source file) of the kernel whose name starts with kernel_prefix, from the committed PMC passes of this workload
    (tools/traffic_pmc*.sh: FETCH_SIZE x the gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc runs) — NOT measured by this run"""
    for rnd in range(PROFILE_ROUND, 0, -1):
        pj = os.path.join(ROOT, "profiles", "round%d_pmc_traffic_%s.json" % (rnd, tag))
        if os.path.exists(pj):
            hits = [v["hbm_bytes_per_launch"] for k, v in json.load(open(pj)).items() if isinstance(v, dict) and k.startswith(kernel_prefix)
                    and "hbm_bytes_per_launch" in v]
            if hits:
                return float(hits[0]), os.path.relpath(pj, ROOT)
    return None, None


def fmlp_kernel_rooflines(lib, _lib, plan, out, T, D, Fh, NL, L, step_s):
    """FMLP (model/fmlp.py:18-39, module/layers.py:740-807): launch durations of the step's heavy kernels (HIP events on the launch stream,
    dr4sr_fmlp_launch_kernel on the state the last step left) -> out["roofline"] for the kernel with the largest share of the step, against
    both roofs; out["roofline_step"]; out["kernel_us_per_step"].  T = B * L: FMLP computes every position of its left-padded rows.
    Algorithmic work per launch (fp32 words, weights and the filter kernel cache-resident):
      ffn_fwd   (k_post_fwd<.., FFN_ONLY>)  2 GEMMs 4 T D F flops; in xf [D]; out a, h [F], u2, z [D], LayerNorm statistics [2]
      ffn_bwd   (k_post_bwd<.., FFN_ONLY>)  2 GEMMs 4 T D F flops; in dz, u2 [D], a [F], statistics; out df, dxf [D], da [F]
      wgrad     (k_fmlp_wgrad_bf64)         2 GEMMs per layer 4 T D F flops; operands da, h [F], df, xf [D] per layer
      filter_fwd / _bwd                     circular convolution along the sequence on the VALU: 2 T L D flops (x 2 backward: dx and dm);
                                            in x [D]; out uf, xf [D] (backward: in dxf, x, uf; out dx)"""
    def launch(kid, layer):
        _lib.check(lib.dr4sr_fmlp_launch_kernel(C.byref(plan), kid, layer, _lib.cur_stream()), "fmlp_launch_kernel")
    work = {"ffn_fwd": (4.0 * T * D * Fh, 4.0 * T * (3 * D + 2 * Fh + 2), NL, True),
            "ffn_bwd": (4.0 * T * D * Fh, 4.0 * T * (4 * D + 2 * Fh + 2), NL, True),
            "wgrad": (4.0 * T * D * Fh * NL, 4.0 * T * (2 * D + 2 * Fh) * NL, 1, True),
            "filter_fwd": (2.0 * T * L * D, 4.0 * T * 3 * D, NL, False),
            "filter_bwd": (4.0 * T * L * D, 4.0 * T * 4 * D, NL, False)}
    ktime, reps = {}, 30
    for name in work:
        kid, tot = _lib.FMLP_KERNEL_IDS[name], 0.0
        layers = range(NL) if work[name][2] > 1 else (0,)
        for layer in layers:
            for _ in range(3):
                launch(kid, layer)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                launch(kid, layer)
            b.record()
            b.synchronize()
            tot += a.elapsed_time(b) * 1e3 / reps
        ktime[name] = tot / len(layers)                      # us per launch, mean over the layers
    step_us = {k: v * work[k][2] for k, v in ktime.items()}
    dom = max(step_us, key=step_us.get)
    fl, by, nl_, mfma = work[dom]
    tf, gbs = fl / (ktime[dom] * 1e-6) / 1e12, by / (ktime[dom] * 1e-6) / 1e9
    kname = {"ffn_fwd": "k_post_fwd<32, 64, 256, true>", "ffn_bwd": "k_post_bwd<32, 64, 256, true>", "wgrad": "k_fmlp_wgrad_bf64",
             "filter_fwd": "k_fmlp_filter_fwd", "filter_bwd": "k_fmlp_filter_bwd"}[dom]
    hbm_bound = gbs / HBM_PEAK_GBS >= tf / MFMA_F32_PEAK_TF or not mfma
    traffic, traffic_src = pmc_traffic("fmlp_B256", kname.split("(")[0]) if T == 256 * L else (None, None)
    out["roofline"] = {"kernel": kname, "bound": "hbm" if hbm_bound else "mfma",
                       "achieved": gbs if hbm_bound else tf, "peak": HBM_PEAK_GBS if hbm_bound else MFMA_F32_PEAK_TF,
                       "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": gbs / HBM_PEAK_GBS if hbm_bound else tf / MFMA_F32_PEAK_TF,
                       "hbm_frac": gbs / HBM_PEAK_GBS, "mfma_f32_frac": tf / MFMA_F32_PEAK_TF if mfma else None,
                       "traffic": traffic, "traffic_source": traffic_src, "us_per_launch": ktime[dom], "launches_per_step": nl_,
                       "flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                       "note": "the flops are the fp32 problem's; the GEMMs run as a bf16x3 split (3 matrix instructions per product) and are "
                               "priced against the fp32 MFMA peak, like the SASRec kernels"}
    tot_fl = 3.0 * NL * 4.0 * T * D * Fh + 3.0 * NL * 2.0 * T * L * D
    out["roofline_step"] = {"bound": "mfma", "achieved": tot_fl / step_s / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                            "frac": tot_fl / step_s / 1e12 / MFMA_F32_PEAK_TF, "flops_per_step": tot_fl,
                            "note": "3 x (Intermediate GEMMs + the filter's circular convolution) on all B * L positions"}
    out["kernel_us_per_step"] = {k: round(v, 2) for k, v in step_us.items()}
    out["roofline_kernels"] = {k: {"us_per_launch": round(ktime[k], 2), "hbm_frac": round(work[k][1] / (ktime[k] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                   "mfma_f32_frac": round(work[k][0] / (ktime[k] * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4) if work[k][3] else None,
                                   "algorithmic_bytes": work[k][1]} for k in work}


def cpu_baseline_leg(rows_np, N, model_kind, p, interval=30):
    """the reference-equivalent CPU step (oracle/ref_trainer.py) on this box's host cores.  torch's default (all
    cores) is the slowest choice for these microsecond-sized ops on a 128-core host, so 8 / 16 / 32 intra-op threads are probed and
    the best is timed with anomaly detection ON (utils/utils.py:11 sets it globally — the reference's real configuration, reported as
    `value`) and OFF as the second column."""
    from oracle.ref_trainer import time_training
    quick = model_kind == "sasrec"
    kw = {"p": p, "model_kind": model_kind, "interval": interval}
    probes = {}
    for th in (8, 16, 32):
        if th <= (os.cpu_count() or 1):
            probes[th] = time_training(rows_np, N, batch_size=256, warmup=2 if quick else 1, max_steps=6 if quick else 3, max_seconds=6.0,
                                       anomaly=True, threads=th, **kw)["seq_per_s"]
    best = max(probes, key=probes.get) if probes else None
    # MetaModel: the timed window must hold whole outer-loop periods (one Hypergrad.grad per `interval` steps)
    steps_on = 50 if quick else (2 * interval if model_kind == "metamodel" else 20)
    r = time_training(rows_np, N, batch_size=256, warmup=3 if quick else 1, max_steps=steps_on, max_seconds=40.0, anomaly=True, threads=best, **kw)
    r_off = time_training(rows_np, N, batch_size=256, warmup=3 if quick else 1, max_steps=30 if quick else (interval if model_kind == "metamodel" else 10),
                          max_seconds=15.0, anomaly=False, threads=best, **kw)
    what = {"sasrec": "nn.TransformerEncoder, multinomial sampler, per-sample DataLoader, Adam",
            "gru4rec": "torch.nn.GRU(bias=False, 2 x 256) behind Dropout(0.2), multinomial sampler, per-sample DataLoader, Adam(weight_decay 1e-4) "
                       "(model/gru4rec.py:12-34, module/layers.py:117-136)",
            "fmlp": "Embedding + position -> LayerNorm -> Dropout(0.5) -> 2 x (torch.fft.rfft / irfft filter, Intermediate 64 -> 256 -> 64), last "
                    "position as the query, one target + one multinomial negative per left-padded row, Adam (model/fmlp.py:8-39, module/layers.py:740-807)",
            "cl4srec": "SASRec step + two item_random views per step (crop / mask / reorder applied row by row on the host, as "
                       "module/data_augmentation.py:20-95) -> encoder -> mean pooling -> InfoNCE batch_both, cl_weight 0.1 (model/cl4srec.py:49-73)",
            "metamodel": "SASRec sub-model + gumbel-softmax selection MLP weighting every step, Hypergrad.grad (double backward, 3 Neumann terms) "
                         "+ clip + SGD every %d steps (model/metamodel.py:95-194, utils/utils.py:134-252); %d outer steps fell into the window"
                         % (interval, r.get("outer_steps", 0))}[model_kind]
    return {"value": r["seq_per_s"], "unit": "sequences/s", "cores": r["threads"], "kind": "port",
            "anomaly_off_value": r_off["seq_per_s"], "thread_probe_seq_per_s": {str(k): v for k, v in probes.items()},
            "sample": "%d steps of B=256 (%.1f s) of oracle/ref_trainer.py: the reference's torch op sequence (%s) on the same synthetic "
                      "rows, anomaly detection ON as utils/utils.py:11 (`anomaly_off_value`: %d steps with it off); %d intra-op threads = "
                      "the best of a 8/16/32 probe; host has %d logical CPUs"
                      % (r["steps"], r["seconds"], what, r_off["steps"], r["threads"], os.cpu_count() or 0)}


def kernel_bytes(kind, T, D, F, n_layer, in_tile=False):
    """ALGORITHMIC HBM bytes of one launch of the fused step's token-tile kernels: every saved activation / gradient row the launch has
    to read or write once, fp32; table rows and weights are cache-resident and not counted.
    in_tile (attention inside the launch, csrc/attn_tile.h): + q, k, v in and the statistics out (forward; ctx becomes an output),
    + q, k, v, statistics in and dq, dk, dv out (backward; dctx no longer leaves the launch)."""
    if in_tile and kind in ("post_fwd", "post_mid", "post_bwd"):
        extra = {"post_fwd": 3 * D + 4, "post_bwd": (3 * D + 4) + 3 * D - D, "post_mid": (3 * D + 4) + (3 * D + 3 * D - D)}[kind]
        base = kernel_bytes(kind, T, D, F, n_layer)
        return base + 4.0 * extra * T
    per = {"post_fwd": 9 * D + 2 * F + 4,                  # ctx, x in; u1, y, a, h, u2, z, next qkv, LayerNorm statistics out
           "post_bwd": (7 * D + F + 4) + (4 * D + F + 2),  # upper dqkv, du1, u2, a, u1, ctx, statistics in; df, da, du1, dout, dctx, rd out
           "post_mid": (5 * D + F) + (9 * D + 3 * F),      # forward + backward of the last layer around the scorer
           "wgrad_fused": (10 * D + 2 * F) * n_layer,      # six (G, X) operand pairs per layer
           "embqkv_fwd": 4 * D + 3, "qkv_embed_bwd": 5 * D}.get(kind)
    return None if per is None else 4.0 * per * T


def bench_metamodel(args):
    """BASELINE configs[4]: MetaModel (DR4SR+) with sub_model = SASRec on toys-shaped synthetic rows, after warm-up:
    every step = weighted fwd/bwd + Adam, every --interval steps one outer hyper-gradient step (9 extra fwd/bwd)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DR4SR_BENCH_SHARE_GPU=1 + DR4SR_BENCH_BACKEND=gloo: debug knobs that run the N-rank code path on ONE GPU (functional check only)
    dev = torch.device("cuda", 0 if os.environ.get("DR4SR_BENCH_SHARE_GPU") else local_rank)
    torch.cuda.set_device(dev)
    from dr4sr_amd import parallel
    if world > 1:
        parallel.init_distributed(dev)                      # gloo control group + the library's RCCL communicator
    os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    import logging
    logging.getLogger("CDR").setLevel(logging.WARNING)
    from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
    cfg = load_config({"model": "MetaModel", "dataset": "synthetic-toys"})
    cfg["model"]["sub_model"] = "SASRec"
    cfg["model"]["sub_overrides"] = {"model": {"dropout_rate": args.dropout}}
    cfg["train"].update({"device": str(dev), "batch_size": args.batch * world, "interval": args.interval, "warmup_epoch": -1})
    if args.dense:
        cfg["data"]["dense"] = True
    seed_everything(cfg["train"]["seed"])
    ds = prepare_datasets(cfg)
    model = prepare_model(cfg, ds)
    model._init_model(ds[0])
    model.train()
    loader = ds[0].get_loader()
    perm = model._perm(loader)
    nb = len(loader)

    fused = world == 1 and model._fused_ok()
    if fused:
        # the model's own epoch machinery (dr4sr_amd/model/metamodel.py:_fused_meta_epoch) on full batches only: selection, negatives,
        # weighting, backward, Adam and the loss log on the device, several steps per graph, outer loop every `interval` steps
        B = loader.batch_size
        n_full = (loader.n // B) * B
        model._perm_buf = perm[:n_full].clone()
        model._perm_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        model._loss_log = torch.zeros(args.warmup + args.steps + 8, dtype=torch.float32, device=dev)
        group, interval = int(cfg["train"].get("steps_per_graph", 16)), int(cfg["train"]["interval"])

        def run(nsteps):
            i = 0
            while i < nsteps:
                k = max(1, min(group, interval - model.step_counter % interval, nsteps - i))
                model._meta_step_graph(loader.fields, B, k)()
                model.step_counter += k
                i += k
                if model.step_counter % interval == 0:
                    model._outter_loop(0)
            return model._loss_log[int(model._perm_counter) - 1]
    else:
        def run(nsteps):
            for i in range(nsteps):
                loss = model._train_batch(model._local_batch(loader, perm, (model.step_counter + 1) % (nb - 1)), 0)   # skip the ragged last batch
            return loss

    run(args.warmup)
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    loss = run(args.steps)
    torch.cuda.synchronize()
    parallel.barrier()
    wall = time.perf_counter() - t0
    # one outer step alone
    bv = model._local_batch(loader, perm, 0)
    bv["neg_item"] = model._neg_sampling(bv)
    bt = model._local_batch(loader, perm, 1)
    bt["neg_item"] = model._neg_sampling(bt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        model.hypergrad_step(bv, bt)
    torch.cuda.synchronize()
    outer_ms = (time.perf_counter() - t1) / 5 * 1e3
    if world > 1:
        wall = parallel.host_allreduce([wall], "max")[0]
    extra = {}
    if rank == 0 and fused:
        # roofline of the weighted inner step's dominant kernel, measured live: the launches of dr4sr_sasrec_fwd_bwd_weighted re-enqueued
        # one by one on the state the last hyper-gradient probe left in the workspace (batch bt), HIP events on the launch stream
        from dr4sr_amd import _lib
        eng = model.engine
        plan = model.sub_model._batch_plan(bt)
        mw = _lib.MetaWeighting()
        mw.phi = model._phi.params.data_ptr()
        uid = bt["user_id"].contiguous()
        mw.user_id = uid.data_ptr()
        mw.tau = model._tau_eff()
        model._fused_weighted(bt)
        torch.cuda.synchronize()
        sl = bt["seqlen"].clamp(0, eng.L).cpu().numpy()
        sasrec_kernel_rooflines(eng.lib, _lib, plan, mw, extra, args, int(sl.shape[0]), eng.L, eng.D, eng.F, eng.n_layer, int(sl.sum()), sl,
                                int(cfg["train"].get("steps_per_graph", 16)), wall / args.steps, dev)
        extra["roofline_step"]["note"] = ("inner weighted step's flops only (the outer loop's 7 forward+backward and 2 forward evaluations "
                                          "every %d steps are in the time, not in the flops)" % args.interval)
        if not args.no_cpu_baseline and world == 1:
            from dr4sr_amd.data.synthetic import make_rows
            rows_np = make_rows(n_items=eng.n_items, seed=2024, dense=args.dense)
            extra["cpu_baseline"] = cpu_baseline_leg(rows_np, eng.n_items, "metamodel", args.dropout, interval=args.interval)
    if rank == 0:
        B = args.batch
        emit({
            **extra,
            "metric": "training sequences/sec, MetaModel(SASRec) d=64 L=50", "value": world * B * args.steps / wall,
            "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MetaModel (DR4SR+) around SASRec on amazon-toys-shaped synthetic rows (BASELINE configs[4]): weighted "
                                   "inner step every step + hyper-gradient outer step every %d steps, B=%d rows/GPU/step, dropout %.2f"
                                   % (args.interval, B, args.dropout),
                       "global_batch": B * world, "seq_len": 50, "parallelism": "dp%d" % world, "hip_graph": True},
            "outer_step_ms": outer_ms, "final_loss": float(loss)})
    parallel.shutdown()


def bench_cl4srec(args):
    """CL4SRec on toys-shaped synthetic rows: BCE step + two augmented views + InfoNCE, API path (single GPU)."""
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    import logging
    logging.getLogger("CDR").setLevel(logging.WARNING)
    from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
    cfg = load_config({"model": "CL4SRec", "dataset": "synthetic-toys"})
    cfg["model"]["dropout_rate"] = args.dropout
    cfg["train"].update({"device": str(dev), "batch_size": args.batch})
    seed_everything(cfg["train"]["seed"])
    ds = prepare_datasets(cfg)
    model = prepare_model(cfg, ds)
    model._init_model(ds[0])
    model.train()
    batches = []
    for b in ds[0].get_loader():
        if b["user_id"].shape[0] == args.batch:
            batches.append(b)
        if len(batches) == 16:
            break

    graph = model._api_graph_ok() and not args.no_graph
    fused = graph and model._fused_cl_ok()              # round 4: fit()'s epoch form — batch selection, negatives, views and the loss log on
    form = "eager"                                       #  the device, k steps per graph (CL4SRec._fused_cl_epoch)
    if fused:
        loader = ds[0].get_loader()
        n_rows = loader.n
        U = (n_rows // args.batch) * args.batch          # whole batches only: every timed step is a full batch
        k = max(1, min(args.steps_per_graph, args.steps))
        model._perm_buf = torch.randperm(n_rows, device=dev)[:U].contiguous()
        model._perm_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        model._loss_log = torch.zeros(U // args.batch + 1, dtype=torch.float32, device=dev)
        run_k, _ = model._fused_cl_graph(loader.fields, args.batch, k)
        run_1, _ = model._fused_cl_graph(loader.fields, args.batch, 1) if k > 1 else (run_k, None)
        nb = U // args.batch
        done = [0]

        def run_steps(n):
            while n > 0:
                if int(done[0] % nb) + (k if n >= k else 1) > nb:      # (host-side bookkeeping only: the device counter wraps the permutation)
                    model._perm_counter.zero_()
                    done[0] = 0
                if n >= k:
                    run_k(); n -= k; done[0] += k
                else:
                    run_1(); n -= 1; done[0] += 1
        run_steps(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        loss = model._loss_log[max(0, done[0] - 1)]
        form = "fused epoch form, %d steps per HIP graph, no per-step host work" % k
    else:
        def step(i):
            batch = dict(batches[i % len(batches)])
            if graph:                                    # the loop body below, captured once and replayed (BaseModel._api_step_graph)
                return model._api_step_graph(batch)
            return model._api_step_body(batch)
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(i)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        form = "API path replayed as one HIP graph per step" if graph else "API path, eager" 
    extra = {}
    if not args.no_kernel_roofline:
        # roofline of the dominant launch, measured live: the launches of the MAIN pass (the batch's SASRec step, one of the step's three
        # encoder passes; the views run the same kernels on 2B shorter rows) re-enqueued one by one on the state a fwd_bwd of one
        # materialised batch leaves in the workspace, HIP events on the launch stream
        from dr4sr_amd import _lib
        eng = model.engine
        b0 = dict(batches[0])
        b0["neg_item"] = model._neg_sampling(b0)
        plan = model._batch_plan(b0)
        eng.fwd_bwd(plan)
        torch.cuda.synchronize()
        sl = b0["seqlen"].clamp(0, eng.L).cpu().numpy()
        sasrec_kernel_rooflines(eng.lib, _lib, plan, None, extra, args, int(sl.shape[0]), eng.L, eng.D, eng.F, eng.n_layer, int(sl.sum()), sl,
                                max(1, min(args.steps_per_graph, args.steps)), wall / args.steps, dev)
        # the whole step's flops: + the two views' encoder passes (forward + backward) on the lengths one draw gives them
        aug = model.augmentation_model.augmentation
        if hasattr(aug, "begin_step"):
            aug.begin_step()
        (_, li), (_, lj) = aug.two_views(b0["in_" + model.fiid], b0["seqlen"]) if hasattr(aug, "two_views") else (aug(b0["in_" + model.fiid], b0["seqlen"]), aug(b0["in_" + model.fiid], b0["seqlen"]))
        vl = torch.cat([li, lj]).clamp(0, eng.L).cpu().numpy().astype(np.float64)
        D_, F_, NL_ = eng.D, eng.F, eng.n_layer
        view_fl = 3.0 * NL_ * (2 * D_ * 3 * D_ + 2 * D_ * D_ + 4 * D_ * F_) * float(vl.sum()) + 3.5 * NL_ * 2 * 2 * D_ * float((vl * (vl + 1) / 2).sum())
        rs = extra["roofline_step"]
        tot = rs["flops_per_step"] + view_fl
        rs.update({"flops_per_step": tot, "achieved": tot / (wall / args.steps) / 1e12, "frac": tot / (wall / args.steps) / 1e12 / MFMA_F32_PEAK_TF,
                   "note": "main pass + the two views' encoder passes (one draw's lengths: %d + %d tokens); InfoNCE's 2 B x 2 B similarity not counted"
                           % (int(sl.sum()), int(vl.sum()))})
        extra["kernel_us_per_step"]["note"] = "launches of the MAIN pass only (one of three encoder passes per step)"
        extra["roofline"].update({"traffic": None, "traffic_source": None})      # (the committed PMC passes are of the plain SASRec step)
    if not args.no_cpu_baseline:
        from dr4sr_amd.data.synthetic import make_rows
        extra["cpu_baseline"] = cpu_baseline_leg(make_rows(n_items=model.engine.n_items - 1, seed=2024, dense=args.dense), model.engine.n_items - 1,
                                                 "cl4srec", args.dropout)
    emit({
        **extra,
        "metric": "training sequences/sec, CL4SRec d=64 L=50", "value": args.batch * args.steps / wall, "unit": "sequences/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "CL4SRec (item_random augmentation, cl_weight 0.1) on amazon-toys-shaped synthetic rows, B=%d, dropout %.2f: "
                               "three encoder passes + InfoNCE per step, %s" % (args.batch, args.dropout, form),
                   "global_batch": args.batch, "seq_len": 50, "parallelism": "dp1", "hip_graph": bool(graph)},
        "final_loss": float(loss.detach())})


def relaunch_under_torchrun(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one per GPU, RCCL)
    and pass their exit code on; rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    share = bool(os.environ.get("DR4SR_BENCH_SHARE_GPU"))
    have = torch.cuda.device_count()
    if have < n_gpus and not share:
        sys.exit("bench.py --gpus %d: this node exposes %d GPU(s).  (DR4SR_BENCH_SHARE_GPU=1 runs the N-rank code path on one GPU over "
                 "the gloo transport — a functional check, not a measurement.)" % (n_gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    if share:
        env.setdefault("DR4SR_DP_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env, stdout=_REAL_STDOUT))       # (this process's fd 1 already points at stderr: guard_stdout)


_REAL_STDOUT = None


def guard_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio when a communicator is created, and that buffer
    is flushed at process exit — i.e. BEHIND the JSON line (seen: `tail -1` of the default run returned "Librccl path : ...").  So file
    descriptor 1 is pointed at stderr for the rest of the process and the JSON line is written to a duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    disarm_crash_line()
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


_CRASH_ARMED = [False]


def arm_crash_line(out, leg):
    """Abort safety of the ONE line.  Every leg after the headline measurement is extra evidence; none of them may
    cost the line.  Python cannot catch a SIGABRT raised by a foreign thread's uncaught C++ exception, a SIGSEGV inside a driver library or
    the SIGTERM a launcher sends to the surviving ranks once another rank died — so rank 0 keeps a COMPLETE copy of the line as it stands,
    marked with the leg that is about to run, inside libdr4sr_hip.so (dr4sr_crash_line_set, include/dr4sr_hip_hooks.h): its signal handler
    write(2)s that copy to the real stdout and _exit(0)s.  Re-armed before every later leg with the line as completed so far; emit() disarms."""
    if _REAL_STDOUT is None or int(os.environ.get("RANK", "0")) != 0:
        return
    snap = dict(out)
    snap["aborted_during"] = leg
    if "collective_forms" in snap:
        cf = dict(snap["collective_forms"])
        cf.setdefault("in_graph_error", "process killed by a signal during: " + leg)
        snap["collective_forms"] = cf
    from dr4sr_amd import _lib
    _lib.load().dr4sr_crash_line_set(json.dumps(snap).encode(), _REAL_STDOUT.fileno(), 0)
    _CRASH_ARMED[0] = True
    if os.environ.get("DR4SR_BENCH_INJECT_ABORT") == leg:          # test hook (tests/test_gpu_zz_transport.py): die the way a c10d thread did
        _REAL_STDOUT.flush()
        os.abort()


def disarm_crash_line():
    if _CRASH_ARMED[0]:
        from dr4sr_amd import _lib
        _lib.load().dr4sr_crash_line_set(None, 0, 0)
        _CRASH_ARMED[0] = False


def main():
    guard_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256, help="rows per GPU per step (configs/basemodel.yaml batch_size)")
    ap.add_argument("--dense", action="store_true", help="all seqlen = 50 (worst case) instead of the toys histogram")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true", help="cl4srec: skip the per-launch timing of the main pass")
    ap.add_argument("--steps-per-graph", type=int, default=30, help="whole training steps captured per HIP graph (single GPU)")
    ap.add_argument("--no-throughput-mode", action="store_true", help="skip the extra B=8192 run reported as `throughput_mode`")
    ap.add_argument("--strong-global-batch", type=int, nargs="*", default=[8192, 32768, 131072, 262144],
                    help="fixed GLOBAL batch sizes of the strong-scaling runs reported as `strong` (per-rank batch = global / N)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling runs")
    ap.add_argument("--gather-tokens", type=int, default=16 * 1024 * 1024, help="tokens of the K1 gather microbench")
    ap.add_argument("--model", default="sasrec", choices=["sasrec", "gru4rec", "fmlp", "metamodel", "cl4srec"],
                    help="sasrec = BASELINE headline (configs[1]); gru4rec = configs[2] (beauty-sized table, dropout 0.2, wd 1e-4); "
                         "fmlp = per-prefix left-padded rows, all B*L positions computed; metamodel = configs[4] (DR4SR+ around "
                         "SASRec: weighted inner steps + one hyper-gradient outer step every --interval steps)")
    ap.add_argument("--embed-dim", type=int, default=64, choices=[64, 128],
                    help="sasrec: 128 = BASELINE configs[3] shape (yelp: N=20034 items, d=128, FFN 128)")
    ap.add_argument("--repeats", type=int, default=15,
                    help="repetitions of the timed region (each exactly --steps steps): ms_per_step = median, ms_per_step_spread = min / max")
    ap.add_argument("--in-graph-timeout", type=int, default=150,
                    help="data parallel: seconds the in-graph-collective attempt (second form) may take before the line is printed without it")
    ap.add_argument("--comm-init-timeout", type=float, default=240.0,
                    help="seconds the RCCL communicator's bootstrap may take before every rank drops to the staged gloo data plane (flagged in the line)")
    ap.add_argument("--no-deterministic-leg", action="store_true", help="skip the extra run of the step in deterministic mode (deterministic_mode)")
    ap.add_argument("--no-dp-leg", action="store_true",
                    help="single GPU: skip the 1-rank RCCL runs of the data-parallel step forms (strong[].dp_1rank_rccl)")
    ap.add_argument("--dp-leg-gpus", type=int, default=8, help="the GPU count whose per-GPU share of each strong-scaling size the 1-rank RCCL leg runs")
    ap.add_argument("--interval", type=int, default=30, help="metamodel: outer-loop period (configs/metamodel.yaml interval)")
    ap.add_argument("--max-seconds", type=int, default=1500,
                    help="whole-run watchdog: after this many seconds the process sends itself SIGTERM — rank 0 then prints the line as "
                         "completed so far (arm_crash_line) instead of hanging the launcher in a wedged collective")
    args = ap.parse_args()
    if args.max_seconds > 0:
        import signal
        import threading
        wd = threading.Timer(args.max_seconds, lambda: os.kill(os.getpid(), signal.SIGTERM))
        wd.daemon = True
        wd.start()
    if args.model == "metamodel":
        return bench_metamodel(args)
    if args.model == "cl4srec":
        return bench_cl4srec(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DR4SR_BENCH_FORCE_DP=1 (under torch.distributed.run --nproc-per-node 1): take the N-rank code path — process group, split graphs,
    # all-reduce between them — with a single rank; exercises the RCCL + graph-capture interplay on a 1-GPU box
    dp = world > 1 or bool(os.environ.get("DR4SR_BENCH_FORCE_DP"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)                 # does not return
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # DR4SR_BENCH_SHARE_GPU=1 (+ DR4SR_DP_BACKEND=gloo): run the N-rank code path on ONE GPU (functional check only)
    dev = torch.device("cuda", 0 if os.environ.get("DR4SR_BENCH_SHARE_GPU") else local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    from dr4sr_amd import parallel
    if dp:
        # gloo control group + the library's own RCCL communicator (no ProcessGroupNCCL); a bootstrap that does not return in time counts as failed
        parallel.init_distributed(dev, allow_fallback=True, init_timeout=args.comm_init_timeout)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from dr4sr_amd import _lib
    from dr4sr_amd.utils.graphs import capture as graph_capture
    from dr4sr_amd.data.synthetic import TOYS_N_ITEMS, make_rows
    from dr4sr_amd.engine import SasrecEngine
    lib = _lib.load()
    if dp and rank == 0 and _REAL_STDOUT is not None:
        # a multi-rank run whose FIRST collective wedges is killed by the watchdog (--max-seconds) or by the launcher before any measurement
        # exists: leave a diagnosable line (value null, exit code 3) instead of silence; the first arm_crash_line() replaces it
        lib.dr4sr_crash_line_set(json.dumps({"metric": "training sequences/sec", "value": None, "unit": "sequences/s", "n_gpus": args.gpus,
                                             "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                                             "aborted_during": "headline measurement (no measurement completed)",
                                             "transport_fallback": parallel.FALLBACK_REASON}).encode(), _REAL_STDOUT.fileno(), 3)
        _CRASH_ARMED[0] = True
        if os.environ.get("DR4SR_BENCH_INJECT_ABORT") == "headline":       # test hook, as in arm_crash_line
            _REAL_STDOUT.flush()
            os.abort()

    def measure(B_arg, steps, warmup, extras, dp=dp, dp_form="host", repeats=None, dp_flat=False, deterministic=False):
        if deterministic:                          # fixed summation order (train.deterministic): the library reads the switch when a plan is carved
            _lib.set_env("DR4SR_DETERMINISTIC", "1")
            try:
                return measure(B_arg, steps, warmup, extras, dp=dp, dp_form=dp_form, repeats=repeats, dp_flat=dp_flat)
            finally:
                _lib.set_env("DR4SR_DETERMINISTIC", None)
        import gc
        gc.collect()
        torch.cuda.empty_cache()                  # the workspace is sized for B * L tokens (72 GiB at 131 072 rows, 145 GiB at 262 144): hand the previous size's blocks back first
        """one timed run of the training step at B_arg rows per GPU; extras = per-kernel launch times + the K1 gather microbench.
        dp=False under a multi-rank launch: every rank runs the single-GPU step on its own (no collective) — the 1-GPU reference
        of the strong-scaling runs.  dp_form (data parallel): "host" = two graphs around a host-launched all-reduce (the model's
        default, dr4sr_amd/model/basemodel.py:_step_graph), "in_graph" = the RCCL all-reduce captured inside the k-step graph
        (returns None when the capture fails on any rank).  dp_flat: one flat all-reduce where the step would have two gradient buckets
        (the at-scale launch forms) — measured beside the bucketed form on a real multi-GPU node, the faster is reported.  The timed region (exactly `steps` steps between barrier + synchronize
        on both sides, MAX over ranks) is repeated `repeats` times: ms_per_step = the median, ms_per_step_spread = [min, max]."""
        repeats = max(1, int(repeats if repeats is not None else args.repeats))
        world = int(os.environ.get("WORLD_SIZE", "1")) if dp else 1
        rank = int(os.environ.get("RANK", "0")) if dp else 0
        B, L, D, H, F, NL, N = B_arg, 50, 64, 2, 128, 2, TOYS_N_ITEMS
        if args.model == "gru4rec":
            N = 12102                                       # amazon-beauty item count (2.Pretrain_regenerator.py:37-42)
        if args.model == "sasrec" and args.embed_dim == 128:
            D, N = 128, 20034                               # yelp item count (2.Pretrain_regenerator.py:37-42), configs[3]
        rows_np = make_rows(n_items=N, seed=2024, dense=args.dense)
        U = rows_np["seqlen"].shape[0]
        data = {k: torch.from_numpy(rows_np[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
        perm = torch.from_numpy(np.random.default_rng(7).permutation(U)).to(dev)      # same permutation on every rank
        rows_buf = torch.zeros(B, dtype=torch.int64, device=dev)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        negbuf = torch.zeros(B, L, dtype=torch.int64, device=dev)
        if args.model == "sasrec":
            eng = SasrecEngine(N, L, D, H, F, NL, 1e-12, args.dropout, B, dev, seed=2023 + 7919 * rank, lr=1e-3)     # per-rank dropout / negative streams, as model/sasrec.py
            init_params_like_reference(eng, 2023)
            # a1 fused: rows_buf is filled by the step's first kernel from (perm, counter); no separate selection launch
            plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=rows_buf, neg_item=negbuf, sample_neg=True,
                                 perm_sel=(perm, B * world, rank * B, counter))
        elif args.model == "gru4rec":
            from dr4sr_amd.gru_engine import GruEngine
            eng = GruEngine(N, L, D, 256, 2, 0.2, B, dev, seed=2023 + 7919 * rank, lr=1e-3, weight_decay=1e-4)
            g = torch.Generator().manual_seed(2023)
            for k, v in eng.views.items():
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / 16.0 if "gru" in k else 0.02 * torch.randn(v.shape, generator=g))
            eng.views["item_embedding.weight"][0] = 0
            plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=rows_buf, neg_item=negbuf, sample_neg=True,
                                 perm_sel=(perm, B * world, rank * B, counter))      # a1 fused into the step's first kernel, as SASRec
        else:
            from dr4sr_amd.fmlp_engine import FmlpEngine
            sl, hist = data["seqlen"], data["in_item_id"]                                # roll every row to a left-padded prefix
            ar = torch.arange(L, device=dev).view(1, -1)
            shift = (L - sl).view(-1, 1)
            data["in_item_id"] = torch.where(ar >= shift, hist.gather(1, (ar - shift) % L), torch.zeros_like(hist)).contiguous()
            data["item_id"] = data["item_id"].gather(1, (sl - 1).clamp(min=0).view(-1, 1)).squeeze(1).contiguous()
            eng = FmlpEngine(N, L, 64, 256, 2, 1e-12, 0.5, B, dev, seed=2023 + 7919 * rank, lr=1e-3)
            g = torch.Generator().manual_seed(2023)
            for k, v in eng.views.items():
                v.copy_(torch.ones(v.shape) if k.endswith("LayerNorm.weight") else (torch.zeros(v.shape) if k.endswith("bias") else 0.02 * torch.randn(v.shape, generator=g)))
            eng.views["item_embedding.weight"][0] = 0
            negbuf = torch.zeros(B, dtype=torch.int64, device=dev)
            plan = eng.make_plan(data["in_item_id"], data["item_id"], rows=rows_buf, neg_item=negbuf, sample_neg=True,
                                 perm_sel=(perm, B * world, rank * B, counter))      # a1 inside the step's first launch
        stream = torch.cuda.Stream(device=dev)

        def select():
            if args.model in ("sasrec", "gru4rec", "fmlp"):
                return
            _lib.check(lib.dr4sr_select_rows(_lib.ptr(perm), U, _lib.ptr(rows_buf), B, B * world, rank * B, _lib.ptr(counter),
                                             _lib.cur_stream()), "select_rows")

        def reduce_grads():
            parallel.allreduce_flat(eng.grads)             # sum over ranks: grads + {n_valid, loss_sum, poison} tail

        # gradient buckets of the data-parallel step (parallel.dp_backward): at scale the item-table gradient is final one launch before
        # the rest, its all-reduce runs beside that launch and only the 280 KB encoder bucket is exposed; one flat all-reduce in the
        # latency forms (and for GRU4Rec / FMLP).  Every rank has B rows here, so every rank decides alike.
        # (two buckets only inside the captured graph: launched from the host the two-bucket step is four submissions per step, +91 us at
        #  16 384 rows per rank with one RCCL rank against +30 us in the graph — the host form stays flat, as in BaseModel._step_graph)
        buckets = parallel.grad_buckets(eng, B, data["seqlen"], want=2) if (dp and args.model == "sasrec" and not dp_flat and dp_form == "in_graph") else None
        two = buckets is not None and len(buckets) == 2

        def step_eager():
            select()
            if not dp:
                eng.train_step(plan)
            else:
                parallel.dp_backward(eng, plan, False, buckets)
                eng.adam_step(plan)

        collective = None
        with torch.cuda.stream(stream):
            for _ in range(3):
                step_eager()
            stream.synchronize()
            use_graph = not args.no_graph
            group = 1
            run_steps = None

            def finish_steps():                              # (forms that leave a step's optimizer pending redefine it)
                pass
            fuse_prep = args.model == "sasrec"              # the optimizer launch of step j prepares step j + 1 (two-phase prep above 1 024 rows)
            if use_graph and not dp:
                # batch selection runs on the device, so consecutive training steps need no host work at all: `group` whole steps
                # are captured into one graph (a graph launch costs ~8 us of idle GPU between replays at this step size); any K / W
                # is served by that graph plus a one-step graph for the remainder
                group = max(1, min(args.steps_per_graph, steps))

                def capture(n):
                    g = torch.cuda.CUDAGraph()
                    with graph_capture(g, stream=stream):
                        if args.model in ("sasrec", "gru4rec"):
                            eng.train_steps(plan, n)         # one prep per graph; each optimizer launch prepares the next step
                        else:
                            for _ in range(n):
                                select()
                                eng.train_step(plan)
                    return g
                g_all = capture(group)
                g_one = capture(1) if group > 1 else g_all

                def run_steps(n):
                    for _ in range(n // group):
                        g_all.replay()
                    for _ in range(n % group):
                        g_one.replay()
            elif use_graph and dp_form == "in_graph":
                # opt-in data-parallel form (train.dp_graph_allreduce): the RCCL all-reduce(s) captured INSIDE the step graph — k whole DP
                # steps per replay, no host work between backward, collective and optimizer; as on one GPU the optimizer launch of
                # step j prepares step j+1.  Measured SECOND, after the host-launched form's numbers are safe (main()).
                group = max(1, min(args.steps_per_graph, steps))

                def capture_dp(n):
                    g = torch.cuda.CUDAGraph()
                    with graph_capture(g, stream=stream):
                        for j in range(n):
                            select()
                            parallel.dp_backward(eng, plan, fuse_prep and j > 0, buckets)
                            if fuse_prep and j < n - 1:
                                eng.adam_step_prepare_next(plan)
                            else:
                                eng.adam_step(plan)
                    return g
                ok = 1
                try:
                    g_all = capture_dp(group)
                    g_one = capture_dp(1) if group > 1 else g_all
                except Exception as e:      # noqa: BLE001
                    print("bench.py: in-graph all-reduce capture failed on rank %d (%s: %s); host-launched collective instead"
                          % (rank, type(e).__name__, e), file=sys.stderr)
                    ok = 0
                # every rank takes the same form: a capture that failed on ANY rank sends all of them to the host-launched collective
                # (ranks mixing in-graph and host-launched collectives would deadlock the communicator)
                stream.synchronize()
                if parallel.all_ok(bool(ok)):
                    def run_steps(n):
                        for _ in range(n // group):
                            g_all.replay()
                        for _ in range(n % group):
                            g_one.replay()
                    collective = "rccl all-reduce captured in the step graph (%d steps per graph, %s)" % (
                        group, "2 buckets: table beside the last weight-gradient launch, then encoder + tail" if two else "1 flat bucket")
                else:
                    return None                              # the caller reports in_graph: null (capture failed on some rank)
            if run_steps is None and use_graph and dp and args.model == "sasrec":
                # Host-launched collective between graphs, one flat bucket.  The graph that holds the optimizer of step j (which also prepares
                # the next batch) holds the backward of step j + 1 as well: ONE graph launch + one collective per step.
                #   [fwd_bwd] AR ([adam+prep | fwd_bwd_prepared] AR)* ... [adam]
                def graph_of(fn):
                    g = torch.cuda.CUDAGraph()
                    with graph_capture(g, stream=stream):
                        fn()
                    return g
                g_first = graph_of(lambda: eng.fwd_bwd(plan))
                g_mid = graph_of(lambda: (eng.adam_step_prepare_next(plan), eng.fwd_bwd_prepared(plan)))
                g_fin = graph_of(lambda: eng.adam_step(plan))
                pending = [False]                            # a step whose optimizer has not run yet (it rides in the next call's first graph)

                def run_steps(n):
                    for _ in range(n):
                        (g_mid if pending[0] else g_first).replay()
                        reduce_grads()
                        pending[0] = True

                def finish_steps():                          # the last step's optimizer: inside the timed region (main loop below)
                    if pending[0]:
                        g_fin.replay()
                        pending[0] = False
                collective = "%s all-reduce launched by the host between graphs (1 flat bucket)" % parallel.backend_name()
            elif run_steps is None and use_graph and dp:
                g_a, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with graph_capture(g_a, stream=stream):
                    select()
                    eng.fwd_bwd(plan)
                with graph_capture(g_b, stream=stream):
                    eng.adam_step(plan)

                def run_steps(n):
                    for _ in range(n):
                        g_a.replay()
                        reduce_grads()
                        g_b.replay()
                collective = "%s all-reduce launched by the host between two graphs" % parallel.backend_name()
            elif run_steps is None:
                def run_steps(n):
                    for _ in range(n):
                        step_eager()
                collective = ("%s all-reduce, eager" % parallel.backend_name()) if dp else None

            run_steps(warmup)
            finish_steps()
            stream.synchronize()
            walls, gpus = [], []
            for _ in range(repeats):
                torch.cuda.synchronize()
                if dp:
                    parallel.barrier()                       # (control plane: a host barrier; the device is drained just above)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                run_steps(steps)
                finish_steps()                               # (host form: the last step's optimizer launch)
                e1.record()
                torch.cuda.synchronize()
                if dp:
                    parallel.barrier()
                walls.append(time.perf_counter() - t0)
                gpus.append(e0.elapsed_time(e1))
            if dp:                                           # every repetition: the MAX over ranks (one control-plane reduction for all of them)
                walls = parallel.host_allreduce(walls, "max")
            order = sorted(range(repeats), key=lambda i: walls[i])
            mid = order[(repeats - 1) // 2]                  # the median repetition (the lower one of an even count)
            wall, gpu_ms = walls[mid], gpus[mid]
            spread = {"repeats": repeats, "median": 1e3 * wall / steps, "min": 1e3 * min(walls) / steps, "max": 1e3 * max(walls) / steps}
            per_rank_ms, coll_us = None, None
            loss, nvalid = eng.loss_and_count()             # (before the stand-alone collective timing below overwrites the gradient tail)
            if dp:
                # this rank's own GPU time per step (events), gathered over the control plane
                per_rank_ms = [round(x, 5) for x in parallel.host_allgather(gpu_ms / steps)]
                # the collective alone: the same flat buffer all-reduced back to back, HIP events around the host-launched form
                # (every rank enters the same count; the gradient buffer is garbage afterwards — nothing reads it before the next step)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    reduce_grads()
                ea.record()
                for _ in range(20):
                    reduce_grads()
                eb.record()
                eb.synchronize()
                coll_us = ea.elapsed_time(eb) * 1e3 / 20
                eng.grads.zero_()
            T_last = int(eng.state[_lib.STATE_T])
            model_desc = {"sasrec": ("SASRec on yelp-sized synthetic rows (BASELINE configs[3] shape): N=20034, L=50, d=128, 2 layers, 2 heads, "
                                     if D == 128 else
                                     "SASRec on amazon-toys-shaped synthetic rows (BASELINE configs[1]): N=11925, L=50, d=64, 2 layers, 2 heads, ")
                                    + "FFN 128, dropout %.2f" % args.dropout,
                          "gru4rec": "GRU4Rec on amazon-beauty-sized synthetic rows (BASELINE configs[2]): N=12102, L=50, d=64, GRU 2x256 no bias, "
                                     "dropout 0.2, Adam wd 1e-4",
                          "fmlp": "FMLP on toys-shaped synthetic per-prefix left-padded rows: N=11925, L=50, d=64, 2 x (filter + FFN 256), "
                                  "dropout 0.5, all B*L positions computed"}[args.model]

            # the arithmetic type per launch form: fp32 everywhere in the latency regime; at scale the six weight-gradient GEMMs run as
            # a 3-term bf16 split with fp32 accumulation (k_wgrad_bf, error 5e-6 against the fp32 kernel: tests), GRU4Rec's
            # cooperative recurrences likewise (k_gru_fwd_wave / k_gru_bwd_coop_bf)
            dtype = "f32"
            if args.model == "sasrec" and bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & 1) and not os.environ.get("DR4SR_WGRAD_F32"):
                dtype = "f32 (weight-gradient GEMMs as bf16x3 split with fp32 accumulation, err 5e-6)"
            if dtype != "f32" and D == 128 and not os.environ.get("DR4SR_TILE_F32"):
                dtype = "f32 io (tile GEMMs and weight-gradient GEMMs as bf16x3 split with fp32 accumulation, err 5e-6 per product)"   # round 5
            if args.model == "fmlp":
                dtype = "f32 io (Intermediate-block GEMMs and weight-gradient GEMMs as bf16x3 split with fp32 accumulation, err 5e-6 per product)"      # k_fmlp_wgrad_bf64 (round 4), tile GEMMs (round 5)
            if args.model == "gru4rec" and bool(lib.dr4sr_gru4rec_uses_cooperative(min(B, 256), 256)) and B <= 1536:
                dtype = "f32 io, bf16x3 recurrent and weight-gradient GEMMs with fp32 accumulation (err 2e-6 / 5e-6)"
            out = {
                "metric": "training sequences/sec, %s d=%d L=50" % ({"sasrec": "SASRec", "gru4rec": "GRU4Rec", "fmlp": "FMLP"}[args.model], D),
                "value": world * B * steps / wall,
                "unit": "sequences/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": 1e3 * wall / steps, "ms_per_step_spread": spread, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": dtype, "data": "synthetic",
                "config": {"workload": "%s, B=%d rows/GPU/step, %s seqlen" %
                                       (model_desc, B, "all-50 (dense)" if args.dense else "toys histogram (10.9% valid)"),
                           "global_batch": B * world, "seq_len": L, "parallelism": "dp%d" % world,
                           "hip_graph": bool(use_graph), "steps_per_graph": group, "collective": collective},
                "gpu_ms_per_step_events": gpu_ms / steps, "final_loss": loss, "valid_tokens_last_step": T_last,
            }
            if dp:
                out["per_rank_gpu_ms_per_step"] = per_rank_ms
                out["allreduce_us_standalone"] = coll_us
                out["allreduce_bytes"] = int(eng.grads.numel()) * 4

            if rank == 0 and args.model == "gru4rec" and extras:
                gru_kernel_rooflines(lib, _lib, plan, out, T_last, D, 256, 2, wall / steps)
            if rank == 0 and args.model == "fmlp" and extras:
                fmlp_kernel_rooflines(lib, _lib, plan, out, B * L, 64, 256, 2, L, wall / steps)
            if rank == 0 and args.model == "sasrec" and extras:
                # ---- per-kernel launch durations, HIP events on the launch stream, on the state of the last step
                seqlen_last = data["seqlen"][rows_buf].clamp(0, L).cpu().numpy()
                ktime = sasrec_kernel_rooflines(lib, _lib, plan, None, out, args, B, L, D, F, NL, T_last, seqlen_last, group, wall / steps, dev)
                # the gather the STEP runs: k_embqkv_fwd / k_wt_embqkv_fwd (gather + position add + dropout + qkv projection of layer 0)
                # on the packed tokens of the last batch: idx read + x row and qkv row written (the table row comes from L2)
                if "embqkv_fwd" in ktime:
                    sb = float(T_last) * (8 + 4 * D + 12 * D)
                    fr = sb / 1e9 / (ktime["embqkv_fwd"] * 1e-6) / HBM_PEAK_GBS
                    out["roofline_gather_step"] = {"kernel": "k_wt_embqkv_fwd" if (bool(lib.dr4sr_sasrec_at_scale(C.byref(plan)) & 1) and D == 64) else "k_embqkv_fwd",
                                                   "bound": "hbm", "achieved": sb / 1e9 / (ktime["embqkv_fwd"] * 1e-6),
                                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr,
                                                   "us_per_launch": ktime["embqkv_fwd"], "tokens": T_last,
                                                   "algorithmic_bytes_per_token": 8 + 16 * D,
                                                   "note": ("launch-latency bound at this size (%d tokens = %.2f MB)" % (T_last, sb / 1e6)) if fr < 0.1 else
                                                           "the gather fused with layer 0's qkv projection (24.6 kF per token): this, not the K1 microbench "
                                                           "(roofline_gather), is what a training step runs"}
                if extras != "full":
                    return out, rows_np, N
                # ---- K1 gather microbench (HBM roofline of the embedding gather): large launch
                ntok = args.gather_tokens
                Bg = ntok // L
                idx = torch.from_numpy(rows_np["in_item_id"]).to(dev)
                reps_rows = (Bg + U - 1) // U
                idx_big = idx.repeat(reps_rows, 1)[:Bg].contiguous()
                idx_big = torch.where(idx_big == 0, torch.randint(1, N, idx_big.shape, device=dev), idx_big)
                outbuf = torch.empty(Bg, L, D, device=dev)
                E, P = eng.views["item_embedding.weight"], eng.views["query_encoder.position_emb.weight"]
                for _ in range(3):
                    lib.dr4sr_embed_gather_posadd(_lib.ptr(E), _lib.ptr(P), _lib.ptr(idx_big), _lib.ptr(outbuf), Bg, L, D, N, _lib.cur_stream())
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    lib.dr4sr_embed_gather_posadd(_lib.ptr(E), _lib.ptr(P), _lib.ptr(idx_big), _lib.ptr(outbuf), Bg, L, D, N, _lib.cur_stream())
                b.record()
                b.synchronize()
                us = a.elapsed_time(b) * 1e3 / 10
                # Bytes of the K1 gather.  SURVEY §8(d) writes the algorithmic figure as idx + table row + output row = 8 + 4D + 4D per
                # token and notes that the 3 MB table is cache-resident (L2 / MALL; the non-temporal output stores keep it there), so
                # the stream that MUST cross the HBM interface is idx-read + output-write = 8 + 4D per token.  The roofline fraction is
                # taken on that stream (a fraction of the HBM peak cannot count bytes that never touch HBM — round 1 reported 1.36 by
                # counting the table row); the 8 + 8D rate is kept beside it as `rate_incl_cached_table_row`.
                hbm_bytes = Bg * L * (8 + 4 * D)
                rate = hbm_bytes / 1e9 / (us * 1e-6)
                traffic, traffic_src = None, None   # HBM bytes per launch from the PMC passes kept under profiles/ (separate runs)
                for rnd in range(PROFILE_ROUND, 0, -1):
                    pj = os.path.join(ROOT, "profiles", "round%d_gather_pmc.json" % rnd)
                    if os.path.exists(pj) and D == 64:          # the PMC passes were taken on the d=64 kernel
                        pm = json.load(open(pj))
                        traffic = Bg * L * (pm["fetch_bytes_per_token_corrected"] + pm["write_bytes_per_token"])
                        traffic_src = os.path.relpath(pj, ROOT)
                        break
                out["roofline_gather"] = {"kernel": "k_embed_dense<%d>" % D, "bound": "hbm", "achieved": rate,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate / HBM_PEAK_GBS,
                                          "traffic": traffic, "traffic_source": traffic_src, "tokens": Bg * L, "us_per_launch": us,
                                          "algorithmic_bytes_per_token": 8 + 4 * D,
                                          "rate_incl_cached_table_row": Bg * L * (8 + 8 * D) / 1e9 / (us * 1e-6),
                                          "pmc_traffic_rate": None if traffic is None else traffic / 1e9 / (us * 1e-6),
                                          "note": "microbench of the dense gather entry point (dr4sr_embed_gather_posadd); the training step "
                                                  "gathers inside k_embqkv_fwd, see roofline_gather_step"}
                del outbuf, idx_big
                # ---- eval hot loop: full-item scores + top-100 of one 2048-row eval batch (basemodel.py:337-365)
                Be, ke = 2048, 100
                qe = torch.randn(Be, D, device=dev)
                hist_e = torch.randint(0, N, (Be, L), device=dev)
                sc_e = torch.empty(Be, ke, device=dev)
                it_e = torch.empty(Be, ke, dtype=torch.int64, device=dev)
                wsb = int(lib.dr4sr_full_score_topk_workspace_bytes(Be, N))
                ws_e = torch.empty(wsb // 4, device=dev)

                def topk_once():
                    _lib.check(lib.dr4sr_full_score_topk_ws(_lib.ptr(qe), _lib.ptr(E), _lib.ptr(hist_e), _lib.ptr(sc_e), _lib.ptr(it_e), Be, D, N,
                                                            L, ke, _lib.ptr(ws_e), wsb, _lib.cur_stream()), "topk_ws")
                for _ in range(3):
                    topk_once()
                a.record()
                for _ in range(10):
                    topk_once()
                b.record()
                b.synchronize()
                ms_e = a.elapsed_time(b) / 10
                out["eval_topk"] = {"rows": Be, "n_items": N, "k": ke, "ms_per_batch": ms_e, "rows_per_s": Be / (ms_e * 1e-3),
                                    "kernels": "k_score_gemm (MFMA) + k_topk_select"}
                del ws_e

        return out, rows_np, N

    out, rows_np, N = measure(args.batch, args.steps, args.warmup, "full")
    if dp:
        out["collective_forms"] = {"host": out["ms_per_step"], "in_graph": None}
        if parallel.FALLBACK_REASON:
            out["transport_fallback"] = parallel.FALLBACK_REASON
    arm_crash_line(out, "throughput_mode")                   # from here on the line cannot be lost (see arm_crash_line)
    tm = None
    sec_rep = max(1, min(args.repeats, 5))                   # the secondary sizes: fewer repetitions of a longer timed region
    if args.model == "sasrec" and args.batch < 8192 and not args.no_throughput_mode:
        # BASELINE.md §3 asks for B=256 (reference batch size) AND B=8192 (throughput / scaling mode); same data, same step.
        # Under a multi-rank launch this run is the single-GPU step on every rank (no collective): the 1-GPU reference of `strong`.
        tm, _, _ = measure(8192, max(20, min(100, args.steps)), 10, "kernels" if world == 1 else None, dp=False, repeats=sec_rep)
        if rank == 0:
            out["throughput_mode"] = {k: tm[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_spread", "steps", "warmup", "dtype", "config",
                                                          "roofline", "roofline_gather_step", "kernel_us_per_step", "valid_tokens_last_step",
                                                          "roofline_step", "roofline_tile_kernels") if k in tm}
    arm_crash_line(out, "deterministic_mode")
    if args.model in ("sasrec", "fmlp", "gru4rec") and not dp and rank == 0 and not args.no_deterministic_leg:
        # what run-to-run determinism costs at this workload (train.deterministic; the reference sets cudnn.deterministic, utils/utils.py:19):
        # the same step with every reduction in a fixed order — at-scale launch forms + ordered partial sums in the weight-gradient launch
        try:
            dm = measure(args.batch, args.steps, args.warmup, None, dp=False, repeats=sec_rep, deterministic=True)[0]
            out["deterministic_mode"] = {"value": dm["value"], "unit": dm["unit"], "ms_per_step": dm["ms_per_step"],
                                         "cost_frac": dm["ms_per_step"] / out["ms_per_step"] - 1.0,
                                         "note": "train.deterministic / DR4SR_DETERMINISTIC=1: bit-identical parameters run to run (tests/test_gpu_deterministic.py); opt-in"}
        except Exception as e:      # noqa: BLE001
            out["deterministic_mode"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    arm_crash_line(out, "strong")
    strong = []
    if args.model == "sasrec" and not args.no_strong and args.embed_dim == 64 and not args.dense and args.batch < 8192:
        # STRONG scaling (north_star: ">= 6x at 8 GPUs"): a FIXED global batch G split over the N ranks (G / N rows per rank per
        # step), one all-reduce per step; `single_gpu_value` = the same G on ONE GPU measured in this very run (every rank runs it
        # as an independent replica, rank 0's number is taken), so speedup and efficiency need no second invocation.
        for G in args.strong_global_batch:
            if G % world or G // world < 1:
                continue
            try:
                one = tm if (G == 8192 and tm is not None) else measure(G, 20, 5, None, dp=False, repeats=sec_rep)[0]
            except Exception as e:      # noqa: BLE001 — e.g. no room for the 145 GiB workspace of 262 144 rows: the size is skipped on EVERY rank alike
                one = None              #  (every rank measures the same single-GPU size on an identical device)
                if rank == 0:
                    out.setdefault("strong_skipped", []).append({"global_batch": G, "error": "%s: %s" % (type(e).__name__, str(e)[:200])})
            if one is None:
                continue
            st_n = one if world == 1 else measure(G // world, max(20, min(100, args.steps)), 10, None, dp=True, repeats=sec_rep)[0]
            strong.append({"global_batch": G, "per_gpu_batch": G // world, "n_gpus": world, "value": st_n["value"], "unit": "sequences/s",
                           "ms_per_step": st_n["ms_per_step"], "ms_per_step_spread": st_n.get("ms_per_step_spread"), "dtype": st_n["dtype"],
                           "single_gpu_value": one["value"], "single_gpu_ms_per_step": one["ms_per_step"],
                           "speedup": st_n["value"] / one["value"], "efficiency": st_n["value"] / one["value"] / world,
                           "collective": st_n["config"].get("collective"),
                           "per_rank_gpu_ms_per_step": st_n.get("per_rank_gpu_ms_per_step"),
                           "allreduce_us_standalone": st_n.get("allreduce_us_standalone")})
        if rank == 0:
            out["strong"] = strong
    arm_crash_line(out, "cpu_baseline")
    if rank == 0 and not dp and not args.no_cpu_baseline and args.model in ("sasrec", "gru4rec", "fmlp"):
        out["cpu_baseline"] = cpu_baseline_leg(rows_np, N, args.model, 0.2 if args.model == "gru4rec" else args.dropout)

    # ---- single GPU: what the DATA-PARALLEL form of the step costs beyond the single-GPU step, measured with the one RCCL rank a 1-GPU
    # box can host (a process group of one).  For every strong-scaling size G the per-GPU share G / 8 runs (a) as the single-GPU k-step
    # graph, (b) as the DP step with its collective(s) captured in the graph, (c) with host-launched collectives between graphs.
    # collective_exposed_us = (b or c) - (a): launch cuts, stream hand-offs and the 1-rank collective kernels — everything of the DP step
    # that does not shrink with the rank count, EXCEPT the wire time of a real 8-rank all-reduce (nothing on this box can measure that;
    # the table bucket's share of it runs beside the last weight-gradient launch, whose duration is printed next to it).
    if world == 1 and not dp and rank == 0 and args.model == "sasrec" and strong and not args.no_dp_leg:
        arm_crash_line(out, "dp_1rank_rccl")
        try:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ["DR4SR_BENCH_FORCE_DP"] = "1"
            parallel.init_distributed(dev)
            Wd = max(2, args.dp_leg_gpus)
            for ent in strong:
                G = ent["global_batch"]
                if G % Wd:
                    continue
                per = G // Wd
                st, wu = max(20, min(100, args.steps)), 10
                plain = measure(per, st, wu, None, dp=False, repeats=sec_rep)[0]
                host = measure(per, st, wu, None, dp=True, dp_form="host", repeats=sec_rep)[0]
                ig = measure(per, st, wu, None, dp=True, dp_form="in_graph", repeats=sec_rep) if parallel.can_capture() else None
                ig = ig[0] if ig is not None else None
                igf = measure(per, st, wu, None, dp=True, dp_form="in_graph", repeats=sec_rep, dp_flat=True) \
                    if (ig is not None and "2 buckets" in (ig["config"].get("collective") or "")) else None
                igf = igf[0] if igf is not None else None
                best = min([host["ms_per_step"]] + [x["ms_per_step"] for x in (ig, igf) if x is not None])
                ent["dp_1rank_rccl"] = {
                    "assumed_gpus": Wd, "per_gpu_batch": per, "single_gpu_form_ms": plain["ms_per_step"],
                    "dp_host_ms": host["ms_per_step"], "dp_in_graph_ms": None if ig is None else ig["ms_per_step"],
                    "dp_in_graph_flat_ms": None if igf is None else igf["ms_per_step"],
                    "collective_exposed_us": 1e3 * (best - plain["ms_per_step"]),         # fastest form (one rank: the flat in-graph form, whose collective is a no-op)
                    "collective_exposed_us_host": 1e3 * (host["ms_per_step"] - plain["ms_per_step"]),
                    "collective_exposed_us_in_graph": None if ig is None else 1e3 * (ig["ms_per_step"] - plain["ms_per_step"]),
                    "collective": (ig or host)["config"]["collective"], "allreduce_us_standalone_flat_1rank": host.get("allreduce_us_standalone"),
                    "projected_speedup_upper_bound": ent["single_gpu_ms_per_step"] / best,
                    "note": "1 RCCL rank: a 1-rank all-reduce is a no-op for RCCL, so this figure is the launch cut of the two-bucket step (+ host "
                            "launches in the host form); the wire time of a real %d-rank all-reduce is NOT in it (upper bound of the speedup; "
                            "dp_in_graph_flat_ms = the same step with one flat all-reduce, whose whole wire time would be exposed)" % Wd}
        except Exception as e:      # noqa: BLE001 — the leg is extra evidence: never lose the line over it
            out["dp_1rank_rccl_error"] = "%s: %s" % (type(e).__name__, e)
        finally:
            os.environ.pop("DR4SR_BENCH_FORCE_DP", None)
            if dist.is_initialized():
                emit(out)
                parallel.shutdown()
                return

    # ---- data parallel, second form: the RCCL all-reduce captured INSIDE the k-step graph (the model's opt-in, train.dp_graph_allreduce).
    # Everything above ran with the host-launched collective — the form every multi-rank test covers — and `out` is complete; the
    # in-graph form is tried only now, under a watchdog: if it hangs (it has never run with more than one RCCL rank before the first
    # multi-GPU run), rank 0 prints the line it already has with in_graph: null and every rank leaves through os._exit.
    if dp:
        host_ms = out["ms_per_step"]
        forms = out["collective_forms"]
        arm_crash_line(out, "in_graph")
        if parallel.can_capture() and args.model == "sasrec" and not args.no_graph and not os.environ.get("DR4SR_DP_HOST_ALLREDUCE"):
            import threading

            def give_up():
                forms["in_graph_error"] = "watchdog: the in-graph form did not finish within %d s" % args.in_graph_timeout
                if rank == 0:
                    emit(out)
                os._exit(0)
            dog = threading.Timer(args.in_graph_timeout, give_up)
            dog.daemon = True
            dog.start()
            try:
                ig = measure(args.batch, args.steps, args.warmup, None, dp=True, dp_form="in_graph")
                if ig is None:
                    forms["in_graph_error"] = "graph capture of the collective failed on at least one rank"
                else:
                    ig = ig[0]
                    forms["in_graph"] = ig["ms_per_step"]
                    forms["in_graph_spread"] = ig["ms_per_step_spread"]
                    forms["in_graph_collective"] = ig["config"]["collective"]
                    if ig["ms_per_step"] < host_ms:            # both are complete training steps: report the faster, name it
                        for k in ("value", "ms_per_step", "ms_per_step_spread", "gpu_ms_per_step_events", "per_rank_gpu_ms_per_step", "final_loss"):
                            out[k] = ig[k]
                        out["config"]["collective"] = ig["config"]["collective"]
                        out["config"]["steps_per_graph"] = ig["config"]["steps_per_graph"]
                    arm_crash_line(out, "in_graph_strong")
                    for ent in strong:                       # every rank walks the same list in the same order
                        sg = measure(ent["global_batch"] // world, max(20, min(100, args.steps)), 10, None, dp=True, dp_form="in_graph",
                                     repeats=sec_rep)
                        if sg is None:
                            continue
                        sg = sg[0]
                        ent["collective_forms"] = {"host": ent["ms_per_step"], "in_graph": sg["ms_per_step"]}
                        if "2 buckets" in (sg["config"].get("collective") or ""):
                            # the bucketed step pays a launch cut (~3 % of the step, profiles/round5_dp_cost_probe.txt) to hide the table
                            # bucket's wire time; whether that pays depends on the node's all-reduce time: measure the flat form too
                            sf = measure(ent["global_batch"] // world, max(20, min(100, args.steps)), 10, None, dp=True, dp_form="in_graph",
                                         repeats=sec_rep, dp_flat=True)
                            if sf is not None:
                                ent["collective_forms"]["in_graph_flat"] = sf[0]["ms_per_step"]
                                if sf[0]["ms_per_step"] < sg["ms_per_step"]:
                                    sg = sf[0]
                        if sg["ms_per_step"] < ent["ms_per_step"]:
                            one_v = ent["single_gpu_value"]
                            ent.update({"value": sg["value"], "ms_per_step": sg["ms_per_step"], "ms_per_step_spread": sg["ms_per_step_spread"],
                                        "speedup": sg["value"] / one_v, "efficiency": sg["value"] / one_v / world,
                                        "collective": sg["config"]["collective"], "per_rank_gpu_ms_per_step": sg.get("per_rank_gpu_ms_per_step")})
            except Exception as e:      # noqa: BLE001
                forms["in_graph_error"] = "%s: %s" % (type(e).__name__, e)
                dog.cancel()
                if rank == 0:
                    emit(out)
                os._exit(0)                                  # the communicator may be wedged: do not enter another collective
            dog.cancel()
    if rank == 0:
        emit(out)
    if dp:
        parallel.shutdown()


if __name__ == "__main__":
    main()
