"""Which rocprofv3 kernel name is the launch `kind` of the fused SASRec step — exactly ONE name per (kind, launch regime).

bench.py's `roofline.traffic` and tools/design_tables.py's PMC column both look a step launch up in a committed PMC digest
(profiles/round<N>_pmc_traffic_<tag>.json, written by tools/traffic_pmc.py: {kernel name: {hbm_bytes_per_launch, ...}}).  Round 5 matched by
PREFIX over both regimes ("k_post_mid" OR "k_wt_post_mid") and summed the hits; once the deterministic-mode leg ran the at-scale kernels in
the same profiled process, the B = 256 headline's `traffic` became the sum of two different kernels (16.64 + 10.03 MB) and DESIGN's column
read `k_wgrad_det_reduce` as the weight-gradient launch.  Here: the regime decides the name, the name must match up to
its template bracket, and more than one hit is an error, not a sum."""

# launch kind -> (latency-regime kernel, at-scale kernel at d = 64 [wave tiles, csrc/linear_wave.hip], at-scale kernel at d = 128)
STEP_KERNELS = {
    "embqkv_fwd":    ("k_embqkv_fwd", "k_wt_embqkv_fwd", "k_embqkv_fwd"),
    "post_fwd":      ("k_post_fwd", "k_wt_post_fwd", "k_post_fwd"),
    "post_mid":      ("k_post_mid", "k_wt_post_mid", "k_post_mid"),
    "post_bwd":      ("k_post_bwd", "k_wt_post_bwd", "k_post_bwd"),
    "qkv_embed_bwd": (None, "k_wt_qkv_embed_bwd", "k_qkv_embed_bwd"),          # latency regime: plane 0 of the weight-gradient launch
    "wgrad_fused":   ("k_wgrad_blk", "k_wgrad_bf64", "k_wgrad_bf"),
    "wgrad":         ("k_wgrad_blk", "k_wgrad_bf64", "k_wgrad_bf"),
    "adam":          ("k_adam", "k_adam", "k_adam"),
    "attn_fwd":      (None, "k_attn_wave_fwd", "k_attn_wave_fwd"),      # round 6: ONE kernel per direction where the lists (several) would run
    "attn_bwd":      (None, "k_attn_wave_bwd", "k_attn_wave_bwd"),
    "prep":          ("k_prep", "k_prep", "k_prep"),
}


def kernel_of(kind, at_scale, d=64):
    """the kernel-name stem (text before the template bracket) of launch `kind`, or None when the regime has no such launch"""
    ent = STEP_KERNELS.get(kind)
    if ent is None:
        return None
    return ent[0] if not at_scale else (ent[1] if int(d) == 64 else ent[2])


def stem(name):
    """'void tiny::k_attn_tiny_bwd<32>(...)' -> 'k_attn_tiny_bwd'"""
    s = name.split("(")[0].split("<")[0].strip()
    return s.split()[-1].split("::")[-1]


def match(pm, kind, at_scale, d=64):
    """(hbm bytes per launch, kernel name) of launch `kind` in the PMC digest `pm`, or (None, None) when the digest does not hold it.
    Raises ValueError when the digest holds MORE than one kernel of that exact stem (two template instances of one kernel in one profiled
    process: the digest cannot say which one the step ran)."""
    want = kernel_of(kind, at_scale, d)
    if want is None:
        return None, None
    hits = [(k, v["hbm_bytes_per_launch"]) for k, v in (pm or {}).items()
            if isinstance(v, dict) and "hbm_bytes_per_launch" in v and stem(k) == want]
    if not hits:
        return None, None
    if len(hits) > 1:
        raise ValueError("PMC digest holds %d kernels named %s: %s" % (len(hits), want, [k for k, _ in hits]))
    return float(hits[0][1]), hits[0][0]


def match_attention(pm, backward):
    """attention launches at scale are SEVERAL kernels per direction (length-class lists): their summed bytes per layer call"""
    def is_bwd(k):
        return "_bwd" in k or (stem(k).startswith("k_attn_tiny") and k.rstrip(">").endswith("true"))
    hits = [v["hbm_bytes_per_launch"] for k, v in (pm or {}).items()
            if isinstance(v, dict) and "hbm_bytes_per_launch" in v and stem(k).startswith("k_attn") and is_bwd(k) == bool(backward)]
    return (float(sum(hits)) if hits else None)
