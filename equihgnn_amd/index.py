"""Per-batch index structures, built once on the device and shared by every layer and by the
backward pass: the two incidence CSRs (by hyperedge / by node), the pooling CSR and the
k-nearest-neighbour lists.

The reference re-derives all of this implicitly inside every torch_scatter / index call from the
unsorted int64 COO lists (conv.py:90-98,172-177); sorting once per batch is what lets the
aggregations run as atomic-free segmented reductions.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def _small_cloud_knn(pos: torch.Tensor, k: int):
    """Self-excluded neighbour lists of a cloud with fewer than k + 1 points, padded to k slots.  The reference takes
    min(k, N) (fa_former_layer.py:664,697: N - 1 others and the point itself at distance 1e9) or min(k, N - 1)
    (equiformer_layer.py:1317-1323) neighbours there and masks by radius; here the list keeps its k slots -- the kernels'
    fixed shape -- and the slots past the N - 1 real neighbours point at the point itself with key 1e9, which the radius
    mask (key <= valid_radius) removes exactly like the reference's own self slot.  A handful of points: plain tensor ops."""
    n = pos.shape[0]
    p = pos.detach().float()
    d = (p.unsqueeze(1) - p.unsqueeze(0)).norm(dim=-1)
    d.fill_diagonal_(float("inf"))
    val, idx = torch.sort(d, dim=-1, stable=True)                    # (ties: the lower index first, as the kernels)
    own = torch.arange(n, device=pos.device).unsqueeze(1)
    nbr = torch.cat((idx[:, :n - 1], own.expand(n, k - (n - 1))), 1)
    key = torch.cat((val[:, :n - 1], torch.full((n, k - (n - 1)), 1e9, dtype=torch.float32, device=pos.device)), 1)
    return nbr.to(torch.int32).contiguous(), key.contiguous()


import os as _os

COUNTED_KNN = not _os.environ.get("EQH_NO_COUNTED_KNN")   # the kNN counts its lists' entries for the transposed CSR (off: clear + histogram launches)


class HyperIndex:
    """CSR views of one batch's incidence structure.

    by_e : rows = hyperedges, col = node of each incidence     (node -> hyperedge aggregation)
    by_v : rows = nodes,      col = hyperedge of each incidence (hyperedge -> node aggregation)
    pool : rows = molecules, entries = nodes (``batch`` is sorted)
    """

    def __init__(self, vertex: torch.Tensor, edges: torch.Tensor, num_nodes: int, num_hyperedges: int,
                 batch: Optional[torch.Tensor] = None, num_graphs: Optional[int] = None):
        self.N, self.M = int(num_nodes), int(num_hyperedges)
        self.nnz = int(vertex.numel())
        problems = [(edges, vertex, self.M), (vertex, edges, self.N)]
        if batch is not None:
            problems.append((batch, None, int(num_graphs)))
        built = ops.csr_build_batch(problems)   # all of them in three launches
        self.by_e, self.by_v = built[0], built[1]
        self.pool = None
        if batch is not None:
            self.B = int(num_graphs)
            self.pool = built[2]
        # int32 gather indices per incidence (null incidences of a padded batch stay -1: the row-gather kernel
        # reads them as zero rows, so they get a zero row forward and a zero gradient backward; the CSRs never
        # list them), int32 `batch`, and the masks of rows a mean leaves at zero -- one launch (hg_index_aux)
        # (the same launch clears the counters of the neighbour search that usually follows: its lists' histogram is the
        # row lengths of the transposed neighbour CSR, whose build then needs neither a clear nor a histogram launch)
        self._knn_counts = torch.empty(self.N + 2, dtype=torch.int32, device=vertex.device) if COUNTED_KNN else None
        self.v32, self.e32, self.batch32, has_v, has_e = ops.index_aux(vertex, edges, batch, self.N, self.M,
                                                                       self.by_v.rowptr, self.by_e.rowptr,
                                                                       self.by_v, self.by_e, self._knn_counts)
        self.has_v, self.has_e = has_v.unsqueeze(-1), has_e.unsqueeze(-1)
        self._knn = {}
        self._knn_pos = {}      # (k, mode) -> address of the coordinates the search ran on (GraphedTrainStep's index prefetch)
        self._pad = None        # (batch, number of real molecules) of a padded batch, set by from_batch
        self._masks = None
        self._he_pool = None
        self.n_box = None   # int32 device [1]: number of real atoms of a padded batch (box of the kNN grid)

    def reset_lazy(self):
        """Forget what the forward pass derives lazily from the index (pad masks, the hyperedge pooling CSR): an index that
        outlives one pass -- the LIVE index of a GraphedTrainStep with index prefetch -- must derive them again inside every
        captured pass, or a second capture would read tensors of the first one's memory pool."""
        self._masks = None
        self._he_pool = None

    def live_clone(self, batch):
        """(clone, dsts, srcs): a structural copy of this index whose tensors are fresh allocations -- what the captured
        training step reads -- and the tensor pairs one batched copy refreshes it with (``ops.copy_many``; every index tensor
        is int32 or fp32, copied as 4-byte words).  ``batch``: the batch object the clone belongs to (its ``batch`` field backs
        ``pad_masks``).  Tensor identity is preserved: a tensor referenced twice (``entry_w_of`` is the partner's ``rowptr``)
        is cloned once."""
        memo, dsts, srcs = {}, [], []

        def cp(v):
            if torch.is_tensor(v):
                hit = memo.get(id(v))
                if hit is None:
                    assert v.element_size() == 4, "index tensors are int32 / fp32"
                    hit = memo[id(v)] = v.detach().clone(memory_format=torch.contiguous_format)
                    if v.numel():
                        dsts.append(hit)
                        srcs.append(v)
                return hit
            if isinstance(v, ops.CSR):
                return ops.CSR(**{k: cp(x) for k, x in v.__dict__.items()})
            if isinstance(v, tuple):
                return tuple(cp(x) for x in v)
            if isinstance(v, list):
                return [cp(x) for x in v]
            if isinstance(v, dict):
                return {k: cp(x) for k, x in v.items()}
            return v

        out = HyperIndex.__new__(HyperIndex)
        for name, v in self.__dict__.items():
            if name in ("_pad", "_masks", "_he_pool", "_knn_counts"):
                continue
            setattr(out, name, cp(v))
        out._masks = out._he_pool = out._knn_counts = None
        out._pad = None if self._pad is None else (batch.batch, self._pad[1])
        return out, dsts, srcs

    def pad_masks(self):
        """(node, hyperedge, incidence, molecule) [rows, 1] float masks of the REAL rows of a padded static-shape batch
        (batch.pad_batch: one dummy molecule owns the padded atoms / hyperedges, padded incidences are null), or four Nones
        for an unpadded batch -- what BatchNorm inside the MLPs needs to keep the padding out of its training statistics."""
        if self._pad is None:
            return (None, None, None, None)
        if self._masks is None:
            batch, real = self._pad
            f = torch.float32
            self._masks = ((batch < real).to(f).unsqueeze(-1), self.has_e.to(f), (self.v32 >= 0).to(f).unsqueeze(-1),
                           (torch.arange(self.B, device=batch.device) < real).to(f).unsqueeze(-1))
        return self._masks

    def hyperedge_pool(self, n_e: torch.Tensor):
        """CSR of hyperedges per molecule (from ``n_e``; hyperedges are stored molecule by molecule,
        data/utils.py:172-178) for the high-order-hyperedge pooling of mhnn.py:72."""
        if self._he_pool is None:
            b = n_e.shape[0]
            e_batch = torch.repeat_interleave(torch.arange(b, device=n_e.device), n_e, output_size=self.M)
            self._he_pool = (ops.csr_build(e_batch, None, b), e_batch.to(torch.int32))
        return self._he_pool

    @classmethod
    def from_batch(cls, data) -> "HyperIndex":
        cached = getattr(data, "_hyper_index", None)
        if cached is not None:
            return cached
        n = getattr(data, "num_nodes", None) or data.x.shape[0]
        m = getattr(data, "num_hyperedges", None)
        if not m:
            ea = getattr(data, "edge_attr", None)
            m = ea.shape[0] if ea is not None else int(data.edge_index1.max()) + 1
        b = getattr(data, "num_graphs", None) or data.y.shape[0]
        idx = cls(data.edge_index0, data.edge_index1, n, m, data.batch, b)
        real = getattr(data, "num_real_graphs", None)
        if real and real < b and idx.pool is not None:
            # padded batch: the atoms of the first `real` molecules come first; their count stays on the device
            idx.n_box = idx.pool.rowptr[real:real + 1]
            idx._pad = (data.batch, int(real))
        try:
            data._hyper_index = idx
        except Exception:  # a frozen container: just rebuild next time
            pass
        return idx

    def knn(self, pos: torch.Tensor, k: int, mode: int):
        """(nbr int32 [N,k], key fp32 [N,k], CSR of the transposed neighbour graph)."""
        hit = self._knn.get((k, mode))
        if hit is None:
            self._knn_pos[(k, mode)] = (pos.data_ptr(), tuple(pos.shape), pos.dtype)
            counts, counted = self._knn_counts, False
            self._knn_counts = None                   # (cleared once, by the index's own launch: the first search takes it)
            if mode == 1 and pos.shape[0] - 1 < k:
                nbr, key = _small_cloud_knn(pos, k)
            elif counts is not None and pos.shape[0] == self.N:
                nbr, key, counted = ops.knn(pos, k, mode, self.n_box, counts=counts)
            else:
                nbr, key = ops.knn(pos, k, mode, self.n_box)
            csr_t = ops.csr_build(nbr.reshape(-1), None, self.N, counts=counts if counted else None)   # int32 keys: no widening copy
            hit = (nbr, key, csr_t)
            self._knn[(k, mode)] = hit
        return hit
