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


def write_vocabulary_text(path, k, L, desc, weight, parent, scoring=0, weighting=0, trailing_newline=False):
    """Write the DBoW2 text format (TemplatedVocabulary::saveToTextFile layout,
    TemplatedVocabulary.h:1424-1447): header `k L scoring weighting`, then per node (file order,
    root omitted) `parent isLeaf d0 .. d31 weight`.  No trailing newline by default: the
    reference loader turns one into a bogus node."""
    n = len(parent)
    nchild = np.bincount(np.asarray(parent[1:], np.int64), minlength=n)
    lines = ["%d %d %d %d" % (k, L, scoring, weighting)]
    for i in range(1, n):
        lines.append("%d %d %s %s" % (parent[i], 1 if nchild[i] == 0 else 0,
                                      " ".join(str(int(b)) for b in desc[i]), repr(float(weight[i]))))
    with open(path, "w") as f:
        f.write("\n".join(lines))
        if trailing_newline:
            f.write("\n")


class ORBVocabulary:
    """Mirror of ORB_SLAM2::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>
    (thirdparty/orb-slam2/include/ORBVocabulary.h:29-34): load from the text format or from a
    packed blob, upload to an extractor context, transform descriptors."""

    def __init__(self, text_file=None, blob=None, cache=False):
        """cache=True: through the binary cache `<text_file>.pgvoc` (pgorb_vocab_load_cached); self.from_cache tells which way."""
        import ctypes as C
        from . import _lib
        self._L = _lib.lib()
        h = C.c_void_p()
        self.from_cache = False
        if text_file is not None and cache:
            fc = C.c_int32(0)
            rc = self._L.pgorb_vocab_load_cached(str(text_file).encode(), C.byref(h), C.byref(fc))
            self.from_cache = bool(fc.value)
        elif text_file is not None:
            rc = self._L.pgorb_vocab_load_text(str(text_file).encode(), C.byref(h))
        else:
            blob = np.ascontiguousarray(blob, np.uint8)
            rc = self._L.pgorb_vocab_from_blob(C.c_void_p(blob.ctypes.data), blob.size, C.byref(h))
        if rc != 0:
            raise ValueError("vocabulary loading failure (rc=%d)" % rc)       # CHECK-fails in the reference
        self._h = h
        vals = [C.c_int32() for _ in range(6)]
        self._L.pgorb_vocab_info(h, *[C.byref(x) for x in vals])
        self.k, self.L, self.nnodes, self.nwords, self.scoring, self.weighting = [x.value for x in vals]
        self._ctx = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pgorb_vocab_free(self._h)
            self._h = None

    def blob(self):
        import ctypes as C
        p, n = C.c_void_p(), C.c_int64()
        self._L.pgorb_vocab_blob(self._h, C.byref(p), C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()

    def upload(self, extractor):
        extractor._check(self._L.pgorb_vocab_upload(extractor._h, self._h))
        self._ctx = extractor

    def transform_features(self, descriptors, levelsup):
        """Per-feature (word id, weight, node id at level L-levelsup) -- computed on the GPU."""
        import ctypes as C
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word, weight, node = np.zeros(n, np.uint32), np.zeros(n, np.float64), np.zeros(n, np.uint32)
        e = self._ctx
        e._check(self._L.pgorb_bow_transform(e._h, C.c_void_p(d.ctypes.data), n, levelsup, C.c_void_p(word.ctypes.data),
                                             C.c_void_p(weight.ctypes.data), C.c_void_p(node.ctypes.data)))
        return word, weight, node

    def transform(self, descriptors, levelsup=4):
        """transform(features, BowVector&, FeatureVector&, levelsup): returns
        ((word_ids, values), (node_ids, starts, feature_indices))."""
        word, weight, node = self.transform_features(descriptors, levelsup)
        return bow_vectors(word, weight, node, self.scoring, self.weighting)


def bow_vectors(word, weight, node, scoring=0, weighting=0):
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    n = len(word)
    word = np.ascontiguousarray(word, np.uint32)
    weight = np.ascontiguousarray(weight, np.float64)
    node = np.ascontiguousarray(node, np.uint32)
    bid, bval = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.float64)
    fnode, fstart, ffeat = np.zeros(n + 1, np.uint32), np.zeros(n + 2, np.int32), np.zeros(n + 1, np.uint32)
    nb, nf = C.c_int32(), C.c_int32()
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = L.pgorb_bow_vectors(n, p(word), p(weight), p(node), scoring, weighting, p(bid), p(bval), C.byref(nb),
                             p(fnode), p(fstart), p(ffeat), C.byref(nf))
    if rc != 0:
        raise ValueError("pgorb_bow_vectors rc=%d" % rc)
    nb, nf = nb.value, nf.value
    end = int(fstart[nf]) if n else 0
    return (bid[:nb].copy(), bval[:nb].copy()), (fnode[:nf].copy(), fstart[:nf + 1].copy(), ffeat[:end].copy())


def bow_score_l1(a, b):
    import ctypes as C
    from . import _lib
    (i1, v1), (i2, v2) = a, b
    i1, i2 = np.ascontiguousarray(i1, np.uint32), np.ascontiguousarray(i2, np.uint32)
    v1, v2 = np.ascontiguousarray(v1, np.float64), np.ascontiguousarray(v2, np.float64)
    p = lambda x: C.c_void_p(x.ctypes.data)
    return _lib.lib().pgorb_bow_score_l1(p(i1), p(v1), len(i1), p(i2), p(v2), len(i2))


def synth_vocabulary_fast(k=10, L=6, seed=7):
    """synth_vocabulary's tree, generated level by level with array operations (ORBvoc.txt's size, k = 10, L = 6,
    is 1 111 111 nodes -- far too many for a Python loop per node).  Same structure: breadth-first file order,
    a child's descriptor = its parent's with ~1/8 of the bits flipped, weights on the leaves."""
    rng = np.random.RandomState(seed)
    n = sum(k ** l for l in range(L + 1))
    desc = np.zeros((n, 32), np.uint8)
    parent = np.full(n, -1, np.int32)
    weight = np.zeros(n, np.float64)
    start, cnt = 0, 1
    for level in range(1, L + 1):
        nxt, m = start + cnt, cnt * k
        par = np.repeat(np.arange(start, start + cnt, dtype=np.int32), k)
        parent[nxt:nxt + m] = par
        if level == 1:
            desc[nxt:nxt + m] = rng.randint(0, 256, (m, 32)).astype(np.uint8)
        else:
            flips = (rng.randint(0, 256, (m, 32)) & rng.randint(0, 256, (m, 32)) & rng.randint(0, 256, (m, 32))).astype(np.uint8)
            desc[nxt:nxt + m] = desc[par] ^ flips
        start, cnt = nxt, m
    weight[start:start + cnt] = np.round(rng.uniform(0.5, 12.0, cnt), 6)
    return desc, weight, parent


def write_vocabulary_text_fast(path, k, L, desc, weight, parent, scoring=0, weighting=0):
    """write_vocabulary_text for ORBvoc-sized trees (one numpy.savetxt call; weights with 6 decimals, which is
    what synth_vocabulary* produce).  No trailing newline."""
    import io
    n = len(parent)
    nchild = np.bincount(np.asarray(parent[1:], np.int64), minlength=n)
    tab = np.empty((n - 1, 35), np.float64)
    tab[:, 0] = parent[1:]
    tab[:, 1] = nchild[1:] == 0
    tab[:, 2:34] = desc[1:]
    tab[:, 34] = weight[1:]
    buf = io.StringIO()
    np.savetxt(buf, tab, fmt=["%d"] * 34 + ["%.6f"], delimiter=" ")
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (k, L, scoring, weighting))
        f.write(buf.getvalue().rstrip("\n"))
