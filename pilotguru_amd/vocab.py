"""ORB vocabulary (DBoW2 TemplatedVocabulary<FORB>) as a flat, broadcastable blob.

The reference builds a pointer tree from a text file (thirdparty/DBoW2/DBoW2/
TemplatedVocabulary.h:1337-1420, node layout :297-329) and shares it by raw pointer with
every System (src/optical_trajectories.cc:87-94).  Here the tree is a struct-of-arrays blob
(ids in file order, children of a node contiguous in `children` order) that one rank parses
or synthesises and broadcasts to its peers with a single RCCL broadcast.

Blob layout (little endian), all sections 64-byte aligned:
  header  int32[16]: magic 'PGVC', version, k, L, nnodes, nwords, scoring, weighting, 0...
  desc    uint8 [nnodes][32]
  weight  float64[nnodes]
  parent  int32 [nnodes]
  child0  int32 [nnodes]   index into `children` of the first child
  nchild  int32 [nnodes]
  word    int32 [nnodes]   word id for leaves, -1 otherwise
  children int32[nnodes-1] child node ids, grouped by parent, in file order
"""
import numpy as np

MAGIC = 0x43564750  # 'PGVC'


def _pad64(n):
    return (n + 63) // 64 * 64


def pack_vocabulary(k, L, desc, weight, parent, scoring=0, weighting=0):
    """desc [n,32] u8, weight [n] f64, parent [n] i32 (node 0 = root, parent[0] = -1)."""
    n = len(parent)
    parent = np.asarray(parent, np.int32)
    order = np.argsort(parent[1:], kind="stable") + 1          # children grouped by parent, file order
    nchild = np.bincount(parent[1:], minlength=n).astype(np.int32)
    child0 = np.zeros(n, np.int32)
    child0[1:] = np.cumsum(nchild)[:-1]
    word = np.full(n, -1, np.int32)
    leaves = np.nonzero(nchild == 0)[0]
    leaves = leaves[leaves != 0]
    word[leaves] = np.arange(len(leaves), dtype=np.int32)      # word ids in file order (:1408-1413)
    hdr = np.zeros(16, np.int32)
    hdr[:8] = [MAGIC, 1, k, L, n, len(leaves), scoring, weighting]
    parts = [hdr.tobytes(), np.ascontiguousarray(desc, np.uint8).tobytes(),
             np.ascontiguousarray(weight, np.float64).tobytes(), parent.tobytes(), child0.tobytes(),
             nchild.tobytes(), word.tobytes(), order.astype(np.int32).tobytes()]
    out = bytearray()
    for p in parts:
        out += p
        out += bytes(_pad64(len(out)) - len(out))
    return np.frombuffer(bytes(out), np.uint8)


def unpack_vocabulary(blob):
    blob = np.asarray(blob, np.uint8)
    hdr = blob[:64].view(np.int32)
    if hdr[0] != MAGIC:
        raise ValueError("not a pgorb vocabulary blob")
    k, L, n, nwords = int(hdr[2]), int(hdr[3]), int(hdr[4]), int(hdr[5])
    off = 64
    out = {"k": k, "L": L, "nnodes": n, "nwords": nwords, "scoring": int(hdr[6]), "weighting": int(hdr[7])}
    for name, dt, cnt in (("desc", np.uint8, n * 32), ("weight", np.float64, n), ("parent", np.int32, n),
                          ("child0", np.int32, n), ("nchild", np.int32, n), ("word", np.int32, n),
                          ("children", np.int32, n - 1)):
        nb = cnt * np.dtype(dt).itemsize
        out[name] = blob[off:off + nb].view(dt)
        off = _pad64(off + nb)
    out["desc"] = out["desc"].reshape(n, 32)
    return out


def synth_vocabulary(k=10, L=5, seed=7):
    """Deterministic full k-ary tree of depth L (the real ORBvoc.txt, k=10 L=6, needs network
    access: fetch-vocabulary.sh:5).  A child's descriptor is its parent's with random bits
    flipped, so greedy descent is meaningful.  Returns (desc, weight, parent) in file order
    (the text format lists nodes parent-before-child; here breadth first)."""
    rng = np.random.RandomState(seed)
    n = sum(k ** l for l in range(L + 1))
    desc = np.zeros((n, 32), np.uint8)
    parent = np.full(n, -1, np.int32)
    weight = np.zeros(n, np.float64)
    start, cnt = 0, 1
    for level in range(1, L + 1):
        nxt = start + cnt
        for p in range(start, start + cnt):
            for c in range(k):
                i = nxt + (p - start) * k + c
                parent[i] = p
                flips = rng.randint(0, 256, 32).astype(np.uint8) & rng.randint(0, 256, 32).astype(np.uint8) \
                    & rng.randint(0, 256, 32).astype(np.uint8)
                desc[i] = (rng.randint(0, 256, 32).astype(np.uint8) if level == 1 else desc[p] ^ flips)
        start, cnt = nxt, cnt * k
    leaves = np.arange(start, start + cnt)
    weight[leaves] = np.round(rng.uniform(0.5, 12.0, len(leaves)), 6)
    return desc, weight, parent


def synth_vocabulary_blob(k=10, L=5, seed=7):
    """torch uint8 tensor holding the packed synthetic vocabulary (for the RCCL broadcast)."""
    import torch
    desc, weight, parent = synth_vocabulary(k, L, seed)
    return torch.from_numpy(pack_vocabulary(k, L, desc, weight, parent).copy())
