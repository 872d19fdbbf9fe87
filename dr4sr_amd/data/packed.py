"""Packed on-disk row cache.

The reference stores a split as a pickled python list of rows (`torch.save(list)`, dataset/preprocess_amazon.ipynb cell 20) and
turns it into tensors with six `torch.tensor([row[i] for row in rows])` list comprehensions (data/dataset.py:79-91) — 12 s for
amazon-toys.  This module keeps reading those files unchanged, and writes next to each `<split>.pth` a flat little-endian int32
image `<split>.pth.dr4srpk` on first use that later runs np.memmap straight into the device upload:

    offset 0   : magic  b"DR4SRPK1"
           8   : u32 version (1) | u32 L | u64 n_rows | u32 target_is_vector | u32 label_is_vector
           32  : u64 source file size | u64 source mtime_ns          (staleness check; mismatch => rebuilt)
           64  : int32 user_id[n] | hist[n,L] | target[n,L] or [n] | seqlen[n] | label[n,L] or [n] | domain_id[n,L]

Train rows carry vector targets/labels ([L]); val/test (and FMLP per-prefix train rows) carry scalars.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Tuple

import numpy as np

MAGIC = b"DR4SRPK1"
HEADER = 64
COLS = ("user_id", "in_item_id", "item_id", "seqlen", "label", "domain_id")


def rows_to_arrays(rows: List[list], L: int) -> Dict[str, np.ndarray]:
    """list of reference rows -> int32 arrays (one pass per column through numpy, not per-element tensor construction)"""
    n = len(rows)
    out = {}
    for i, name in enumerate(COLS):
        first = rows[0][i] if n else 0
        if isinstance(first, (list, tuple)):
            a = np.asarray([r[i] for r in rows], dtype=np.int32).reshape(n, -1)
            if a.shape[1] != L:
                raise ValueError(f"column {name}: row length {a.shape[1]} != max_seq_len {L}")
        else:
            a = np.asarray([r[i] for r in rows], dtype=np.int32).reshape(n)
        out[name] = a
    return out


def _layout(n: int, L: int, tvec: bool, lvec: bool) -> List[Tuple[str, tuple]]:
    return [("user_id", (n,)), ("in_item_id", (n, L)), ("item_id", (n, L) if tvec else (n,)), ("seqlen", (n,)),
            ("label", (n, L) if lvec else (n,)), ("domain_id", (n, L))]


def write_packed(path: str, arrays: Dict[str, np.ndarray], L: int, src: str) -> None:
    n = int(arrays["user_id"].shape[0])
    tvec, lvec = arrays["item_id"].ndim == 2, arrays["label"].ndim == 2
    st = os.stat(src)
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        hdr = MAGIC + struct.pack("<IIQII", 1, L, n, int(tvec), int(lvec)) + struct.pack("<QQ", st.st_size, st.st_mtime_ns)
        f.write(hdr.ljust(HEADER, b"\0"))
        for name, shp in _layout(n, L, tvec, lvec):
            a = np.ascontiguousarray(arrays[name], dtype="<i4")
            assert a.shape == shp, (name, a.shape, shp)
            f.write(a.tobytes())
    os.replace(tmp, path)                      # atomic: concurrent ranks either see the old file or the complete new one


def read_packed(path: str, src: str = None, L: int = None) -> Dict[str, np.ndarray]:
    """memory-map a packed image; returns None if it is missing, malformed, stale w.r.t. `src`, or was written for another
    max_seq_len than the `L` asked for (the image is then rebuilt from the .pth, whose rows are checked against L)"""
    try:
        with open(path, "rb") as f:
            hdr = f.read(HEADER)
    except OSError:
        return None
    if len(hdr) < HEADER or hdr[:8] != MAGIC:
        return None
    ver, hdr_L, n, tvec, lvec = struct.unpack("<IIQII", hdr[8:32])
    size, mtime = struct.unpack("<QQ", hdr[32:48])
    if ver != 1 or (L is not None and int(hdr_L) != int(L)):
        return None
    if src is not None:
        try:
            st = os.stat(src)
        except OSError:
            st = None                            # source gone: the packed image is all there is
        if st is not None and (st.st_size != size or st.st_mtime_ns != mtime):
            return None
    lay = _layout(n, hdr_L, bool(tvec), bool(lvec))
    need = HEADER + 4 * sum(int(np.prod(s)) for _, s in lay)
    if os.path.getsize(path) != need:
        return None
    mm = np.memmap(path, dtype="<i4", mode="r", offset=HEADER)
    out, off = {}, 0
    for name, shp in lay:
        k = int(np.prod(shp))
        out[name] = mm[off:off + k].reshape(shp)
        off += k
    return out


def load_split(pth_path: str, L: int, use_cache: bool = True) -> Dict[str, np.ndarray]:
    """arrays of one split: packed image if fresh, else unpickle the reference's .pth (and refresh the image)"""
    pk = pth_path + ".dr4srpk"
    if use_cache:
        got = read_packed(pk, pth_path, L)
        if got is not None:
            return got
    import torch
    rows = torch.load(pth_path, weights_only=False)
    arrays = rows_to_arrays(rows, L)
    if use_cache:
        try:
            write_packed(pk, arrays, L, pth_path)
        except OSError:
            pass                                 # read-only dataset directory: keep working from memory
    return arrays
