"""Commonsense triplet filter (reference ``evaluator.py:189-194,261-266``; ``train_utils.py:49-50``).

The reference tests ``tuple(triplet) in dict`` per candidate row on the host (a device->host copy and a Python
hash lookup per row).  Here the two triplet sets become device bitmaps over the num_classes x num_relations x
num_classes space (150*50*150 bits = 141 KB) and the filter is one HIP kernel over all candidates.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class TripletBitmaps:
    def __init__(self, aligned_keys, violated_keys, num_classes: int, num_relations: int, device):
        self.C, self.R = int(num_classes), int(num_relations)
        self.device = torch.device(device)
        self.aligned = self._pack(aligned_keys)
        self.violated = self._pack(violated_keys)

    def _pack(self, keys) -> torch.Tensor:
        nbits = self.C * self.R * self.C
        nwords = (nbits + 31) // 32
        bits = np.zeros(nwords * 32, dtype=np.uint64)
        for (s, r, o) in keys:
            if 0 <= s < self.C and 0 <= o < self.C and 0 <= r < self.R:
                bits[(s * self.R + r) * self.C + o] = 1
        words = (bits.reshape(nwords, 32) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)   # bit b of word w
        return torch.from_numpy(words.view(np.int32).copy()).to(self.device)

    def filter_(self, subject_cat: torch.Tensor, relation_pred: torch.Tensor, object_cat: torch.Tensor,
                confidence: torch.Tensor) -> torch.Tensor:
        """confidence[i] = -inf where the triplet is violated or not aligned (in place, GPU only)."""
        if not confidence.is_cuda:
            raise RuntimeError("the commonsense filter runs on the GPU (sgc_commonsense_filter); move the inputs to cuda")
        lib = _lib.load()
        s, r, o = (t.to(torch.int64).contiguous() for t in (subject_cat, relation_pred, object_cat))
        assert confidence.dtype == torch.float32 and confidence.is_contiguous()
        _lib.check(lib.sgc_commonsense_filter(_lib.ptr(s), _lib.ptr(r), _lib.ptr(o), _lib.ptr(confidence), int(confidence.numel()),
                                              _lib.ptr(self.aligned), _lib.ptr(self.violated), self.C, self.R,
                                              _lib.stream_ptr()), "sgc_commonsense_filter")
        return confidence
