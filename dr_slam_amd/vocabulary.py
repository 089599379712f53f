"""ORB vocabulary handling (host side of the BoW step).

The reference loads `Vocabulary/ORBvoc.txt` through DBoW2's TemplatedVocabulary::loadFromTextFile
(reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424); that blob is not in the reference
checkout (.MISSING_LARGE_BLOBS), so this module can also synthesise a vocabulary of the same shape
(k-ary tree of depth L over 256-bit descriptors, idf-like weights) and read/write the text format:

    k L scoring weighting
    parent isLeaf d0 d1 ... d31 weight        (one line per node, ids = line number, root = 0)

BowVector / FeatureVector (std::map in DBoW2) are rebuilt on the host from the per-feature device
results in feature order, because addWeight's float64 accumulation depends on that order.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .synth import splitmix64

L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)
TF_IDF, TF, IDF, BINARY = range(4)


@dataclass
class Vocabulary:
    k: int
    L: int
    scoring: int
    weighting: int
    parent: np.ndarray    # int32 [n], parent[0] unused
    is_leaf: np.ndarray   # uint8 [n]
    desc: np.ndarray      # uint8 [n, 32]
    weight: np.ndarray    # float64 [n]

    @property
    def n_nodes(self):
        return len(self.parent)

    def to_text(self) -> str:
        lines = [f"{self.k} {self.L} {self.scoring} {self.weighting}"]
        for i in range(1, self.n_nodes):
            d = " ".join(str(int(v)) for v in self.desc[i])
            lines.append(f"{int(self.parent[i])} {int(self.is_leaf[i])} {d} {float(self.weight[i])!r}")
        return "\n".join(lines) + "\n"

    @staticmethod
    def from_text(text: str) -> "Vocabulary":
        rows = text.strip().split("\n")
        k, L, n1, n2 = (int(v) for v in rows[0].split()[:4])
        if k < 0 or k > 20 or L < 1 or L > 10 or n1 < 0 or n1 > 5 or n2 < 0 or n2 > 3:
            raise ValueError("Vocabulary loading failure: This is not a correct text file!")
        n = len(rows)
        parent = np.zeros(n, np.int32)
        leaf = np.zeros(n, np.uint8)
        desc = np.zeros((n, 32), np.uint8)
        weight = np.zeros(n, np.float64)
        for i, r in enumerate(rows[1:], start=1):
            t = r.split()
            parent[i], leaf[i] = int(t[0]), int(t[1]) > 0
            desc[i] = [int(v) for v in t[2:34]]
            weight[i] = float(t[34])
        return Vocabulary(k, L, n1, n2, parent, leaf, desc, weight)

    def upload(self, ctx):
        ctx.voc_upload(self.k, self.L, self.scoring, self.weighting, self.parent, self.desc, self.weight, self.is_leaf)

    def pack(self) -> np.ndarray:
        """Flat byte blob (what rank 0 broadcasts over RCCL in the multi-GPU mode)."""
        head = np.array([self.k, self.L, self.scoring, self.weighting, self.n_nodes, 0, 0, 0], np.int32)
        return np.concatenate([head.view(np.uint8), self.parent.view(np.uint8), self.weight.view(np.uint8),
                               self.is_leaf, self.desc.reshape(-1)])

    @staticmethod
    def unpack(blob: np.ndarray) -> "Vocabulary":
        blob = np.ascontiguousarray(blob, np.uint8)
        k, L, sc, we, n = (int(v) for v in blob[:32].view(np.int32)[:5])
        o = 32
        parent = blob[o:o + 4 * n].view(np.int32).copy(); o += 4 * n
        weight = blob[o:o + 8 * n].view(np.float64).copy(); o += 8 * n
        leaf = blob[o:o + n].copy(); o += n
        desc = blob[o:o + 32 * n].reshape(n, 32).copy()
        return Vocabulary(k, L, sc, we, parent, leaf, desc, weight)


def make_synthetic(k: int = 10, L: int = 6, seed: int = 1, scoring: int = L1_NORM, weighting: int = TF_IDF,
                   stop_fraction: float = 0.01) -> Vocabulary:
    """Full k-ary tree of depth L: children are created consecutively per parent (as DBoW2's HKmeansStep
    does), each child descriptor = parent descriptor with ~256/(2^(level+1)) random bits flipped, leaf
    weights are idf-like positives with `stop_fraction` of words stopped (weight 0)."""
    n = (k ** (L + 1) - 1) // (k - 1)
    parent = np.zeros(n, np.int32)
    level = np.zeros(n, np.int32)
    start, cnt = 1, k
    prev_start = 0
    for lvl in range(1, L + 1):
        ids = np.arange(start, start + cnt)
        parent[ids] = prev_start + (ids - start) // k
        level[ids] = lvl
        prev_start, start, cnt = start, start + cnt, cnt * k
    is_leaf = (level == L).astype(np.uint8)
    desc = np.zeros((n, 32), np.uint8)
    order = np.arange(1, n)
    h = splitmix64(order.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed))
    for lvl in range(1, L + 1):
        ids = np.nonzero(level == lvl)[0]
        base = desc[parent[ids]]
        nflip = max(2, 128 >> lvl)
        for j in range(nflip):
            hv = splitmix64(h[ids - 1] + np.uint64(j * 7919 + lvl))
            bit = (hv % np.uint64(256)).astype(np.int64)
            base[np.arange(len(ids)), bit >> 3] ^= (1 << (bit & 7)).astype(np.uint8)
        desc[ids] = base
    u = (splitmix64(h + np.uint64(12345)) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    weight = np.zeros(n, np.float64)
    weight[1:] = np.where(is_leaf[1:] > 0, 0.5 + 8.0 * u, 0.0)
    stopped = (splitmix64(h + np.uint64(777)) % np.uint64(10000)).astype(np.float64) < stop_fraction * 10000
    weight[1:][(is_leaf[1:] > 0) & stopped] = 0.0
    return Vocabulary(k, L, scoring, weighting, parent, is_leaf, desc, weight)


def bow_and_feature_vectors(voc: Vocabulary, word: np.ndarray, weight: np.ndarray, nid: np.ndarray):
    """TemplatedVocabulary::transform(features, v, fv, levelsup) container part (:1127-1190):
    returns (BowVector as sorted ids + values, FeatureVector as {node: [feature idx, ...]})."""
    must = voc.scoring in (L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA)
    bow: dict[int, float] = {}
    fv: dict[int, list[int]] = {}
    tf = voc.weighting in (TF_IDF, TF)
    for i in range(len(word)):
        w = float(weight[i])
        if w > 0:
            wid = int(word[i])
            if tf:
                bow[wid] = bow[wid] + w if wid in bow else w      # BowVector::addWeight
            elif wid not in bow:
                bow[wid] = w                                      # addIfNotExist
            fv.setdefault(int(nid[i]), []).append(i)
    ids = sorted(bow)
    vals = [bow[i] for i in ids]
    if tf and vals and not must:
        nd = float(len(vals))
        vals = [v / nd for v in vals]
    if must:
        if voc.scoring == L2_NORM:
            norm = 0.0
            for v in vals:
                norm += v * v
            norm = float(np.sqrt(norm))
        else:
            norm = 0.0
            for v in vals:
                norm += abs(v)
        if norm > 0.0:
            vals = [v / norm for v in vals]
    return np.array(ids, np.int32), np.array(vals, np.float64), dict(sorted(fv.items()))
