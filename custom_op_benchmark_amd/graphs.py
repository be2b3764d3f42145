"""Graph containers and synthetic graph builders (host-side plumbing, torch tensors).

The reference keeps a graph as eight loose int64 tensors: the row-major chunked CSR
``(ROW, INDPTR_R, eid_r, indices_r)`` and the column-major one ``(COL, INDPTR_C, eid_c,
indices_c)`` (``wrapper.py:84-112,198-199``).  ``AttnGraph`` bundles them in the
positional order the reference's ``Function.apply`` calls use (``wrapper.py:22,46``).

Builders generalise the reference fixture construction (``wrapper.py:93-112``): row-major
CSR sorted by (src, dst) with ``eid_r = arange(E)``; column-major CSR by a stable sort on
dst whose ``eid_c`` maps column-order slots back to row-order edge ids.
"""
from dataclasses import dataclass

import torch

from .part_csr import partition_csr


@dataclass
class AttnGraph:
    n_src: int            # rows of the adjacency (query / destination-of-aggregation nodes)
    n_dst: int            # columns (key / value nodes)
    n_edges: int
    src: torch.Tensor     # (E) row id per edge, row-major order
    dst: torch.Tensor     # (E) col id per edge, row-major order
    indptr_r: torch.Tensor
    indptr_c: torch.Tensor
    row: torch.Tensor     # ROW      chunk -> row id       (partition_csr of indptr_r)
    ptr_r: torch.Tensor   # INDPTR_R chunk -> first slot
    eid_r: torch.Tensor
    indices_r: torch.Tensor
    col: torch.Tensor     # COL, INDPTR_C, eid_c, indices_c: same for the transposed CSR
    ptr_c: torch.Tensor
    eid_c: torch.Tensor
    indices_c: torch.Tensor
    chunk_size: int = 32

    def csr_args(self):
        """The 8 leading positional args of MaskedMMCSR.apply / VectorSPMM.apply."""
        return (self.row, self.ptr_r, self.eid_r, self.indices_r,
                self.col, self.ptr_c, self.eid_c, self.indices_c)

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return AttnGraph(**kw)

    @property
    def n_row_chunks(self):
        return int(self.row.numel())

    @property
    def n_col_chunks(self):
        return int(self.col.numel())


def graph_from_coo(src, dst, n_src, n_dst=None, chunk_size=32, presorted=False):
    """Build both chunked CSR orientations from an edge list (duplicates are kept: the
    kernels treat them as distinct edges)."""
    n_dst = n_src if n_dst is None else n_dst
    src = src.to(torch.int64)
    dst = dst.to(torch.int64)
    E = int(src.numel())
    dev = src.device
    if not presorted and E:
        key = src * n_dst + dst
        order = torch.argsort(key, stable=True)
        src, dst = src[order], dst[order]
        del key, order
    indptr_r = torch.zeros(n_src + 1, dtype=torch.int64, device=dev)
    indptr_c = torch.zeros(n_dst + 1, dtype=torch.int64, device=dev)
    if E:
        indptr_r[1:] = torch.cumsum(torch.bincount(src, minlength=n_src), 0)
        indptr_c[1:] = torch.cumsum(torch.bincount(dst, minlength=n_dst), 0)
    eid_r = torch.arange(E, dtype=torch.int64, device=dev)          # wrapper.py:100
    indices_r = dst
    # stable sort by dst keeps src ascending inside a column: same order as wrapper.py:104-112
    eid_c = torch.argsort(dst, stable=True) if E else eid_r.clone()
    indices_c = src[eid_c]
    row, ptr_r = partition_csr(indptr_r, chunk_size)
    col, ptr_c = partition_csr(indptr_c, chunk_size)
    return AttnGraph(n_src, n_dst, E, src, dst, indptr_r, indptr_c, row, ptr_r, eid_r, indices_r,
                     col, ptr_c, eid_c, indices_c, chunk_size)


def block_diagonal_graph(batch_size, l, chunk_size=32, device="cpu"):
    """The reference harness fixture: ``batch_size`` disjoint complete digraphs with
    self-loops on ``l`` nodes each (``wrapper.py:79-112``).  n = bs*l, e = bs*l*l."""
    b = torch.arange(batch_size, device=device).view(-1, 1, 1)
    x = torch.arange(l, device=device).view(1, -1, 1)
    y = torch.arange(l, device=device).view(1, 1, -1)
    src = (b * l + x).expand(batch_size, l, l).reshape(-1)
    dst = (b * l + y).expand(batch_size, l, l).reshape(-1)
    return graph_from_coo(src, dst, batch_size * l, chunk_size=chunk_size, presorted=True)


def uniform_random_graph(n_nodes, n_edges, seed=0, chunk_size=32, device="cpu", n_dst=None):
    g = torch.Generator(device=device).manual_seed(seed)
    n_dst = n_nodes if n_dst is None else n_dst
    src = torch.randint(0, n_nodes, (n_edges,), generator=g, device=device)
    dst = torch.randint(0, n_dst, (n_edges,), generator=g, device=device)
    return graph_from_coo(src, dst, n_nodes, n_dst, chunk_size)


def powerlaw_weights(n_nodes, alpha, seed, device, shuffle=True):
    """Chung-Lu node weights w_i ~ (rank_i + r0)^-alpha with node ids shuffled, so hubs
    are not clustered by id (shuffle=False: ids sorted by degree, hubs first).  alpha=0 is uniform."""
    g = torch.Generator(device=device).manual_seed(seed)
    rank = torch.arange(n_nodes, dtype=torch.float64, device=device)
    w = (rank + 10.0).pow(-alpha)
    perm = torch.randperm(n_nodes, generator=g, device=device)
    if not shuffle:
        return w / w.sum()
    out = torch.empty_like(w)
    out[perm] = w
    return out / out.sum()


LABELINGS = ("shuffled", "degree", "clustered")


def chung_lu_graph(n_nodes, n_edges, alpha=0.5, seed=0, chunk_size=32, device="cpu",
                   batch=1 << 26, labeling="shuffled", community=1024, p_in=0.9):
    """Reddit-shape stand-in: both endpoints of every edge drawn from one power-law node
    weight vector (expected degree of node i = E * w_i on both sides), sampled on
    ``device`` by inverse-CDF search in batches.
    labeling (how node ids relate to the structure; the kernels must not care -- bench.py --labeling):
      "shuffled"   hubs spread uniformly over the ids (the default);
      "degree"     ids sorted by expected degree, hubs first (what a degree-ordered real graph looks like);
      "clustered"  stochastic-block communities of `community` consecutive ids: an edge's destination lies in its
                   source's community with probability p_in (drawn by weight inside it), anywhere otherwise --
                   the generalisation of the reference's own fixture (wrapper.py:84-112: disjoint complete blocks,
                   i.e. p_in = 1 at community = 30); hubs are spread over the communities."""
    if labeling not in LABELINGS:
        raise ValueError("labeling must be one of %s" % (LABELINGS,))
    w = powerlaw_weights(n_nodes, alpha, seed, device, shuffle=(labeling != "degree"))
    cdf = torch.cumsum(w, 0)
    cdf[-1] = 1.0
    g = torch.Generator(device=device).manual_seed(seed + 1)
    srcs, dsts = [], []
    for s in range(0, n_edges, batch):
        m = min(batch, n_edges - s)
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        src = torch.searchsorted(cdf, u).clamp_(max=n_nodes - 1)
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        dst = torch.searchsorted(cdf, u).clamp_(max=n_nodes - 1)
        if labeling == "clustered":
            lo = (src // community) * community
            hi = (lo + community).clamp_(max=n_nodes)
            c_lo = torch.where(lo > 0, cdf[(lo - 1).clamp_(min=0)], torch.zeros((), dtype=cdf.dtype, device=device))
            c_hi = cdf[hi - 1]
            u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
            inside = torch.searchsorted(cdf, c_lo + u * (c_hi - c_lo)).clamp_(max=n_nodes - 1)
            inside = torch.minimum(torch.maximum(inside, lo), hi - 1)
            keep = torch.rand(m, generator=g, device=device) < p_in
            dst = torch.where(keep, inside, dst)
        srcs.append(src); dsts.append(dst)
    src, dst = torch.cat(srcs), torch.cat(dsts)
    del srcs, dsts
    return graph_from_coo(src, dst, n_nodes, n_nodes, chunk_size)


def rmat_edges(scale, n_edges, seed=0, device="cpu", abcd=(0.57, 0.19, 0.19, 0.05), src_prefix_bits=0,
               src_prefix=0, batch=1 << 25):
    """R-MAT edge list on 2**scale nodes (Graph500 parameters by default, no vertex permutation:
    low ids are the hubs).  With src_prefix_bits > 0 the top bits of every source are fixed to
    `src_prefix` and the matching destination bits are drawn from the conditional quadrant
    probabilities: the edges of ONE node-range shard, generated on that shard's device."""
    a, b, c, d = abcd
    g = torch.Generator(device=device).manual_seed(seed)
    srcs, dsts = [], []
    for s0 in range(0, n_edges, batch):
        m = min(batch, n_edges - s0)
        src = torch.zeros(m, dtype=torch.int64, device=device)
        dst = torch.zeros(m, dtype=torch.int64, device=device)
        for level in range(scale):
            u = torch.rand(m, generator=g, device=device)
            if level < src_prefix_bits:
                sb = (src_prefix >> (src_prefix_bits - 1 - level)) & 1
                p_d1 = (b / (a + b)) if sb == 0 else (d / (c + d))
                sbit = torch.full((m,), sb, dtype=torch.int64, device=device)
                dbit = (u < p_d1).to(torch.int64)
            else:
                # quadrants in order a (0,0), b (0,1), c (1,0), d (1,1)
                sbit = (u >= a + b).to(torch.int64)
                dbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
            src = (src << 1) | sbit
            dst = (dst << 1) | dbit
        srcs.append(src); dsts.append(dst)
    return torch.cat(srcs), torch.cat(dsts)


SHAPES = {
    # name: (nodes, edges)  -- shape-matched synthetic stand-ins (no datasets in the image)
    "cora": (2708, 10556),
    "reddit": (232965, 114615892),
    "products": (2449029, 61859140),
    "harness": (15360, 460800),
    # BASELINE.json configs 4 and 5: whole-graph sizes; bench.py runs ONE 1/8 shard per GPU
    "papers100m": (111059956, 1615685872),
    "rmat25": (1 << 25, 1 << 30),
}
SHARDS_OF = {"papers100m": 8, "rmat25": 8}      # the partition the multi-GPU configs are quoted on
DEFAULT_D = {"papers100m": 128, "rmat25": 256}  # per BASELINE.json


# ---- plans: explicit lifecycle ---------------------------------------------------------------------
def prepare(g, h=1, d=64, dtype=torch.float32, fused=True):
    """Build the per-graph plans of both orientations AND every window structure the operators will
    use for node tensors of (n, h, d) `dtype` values, now.  Afterwards no op call on this graph and
    these shapes allocates or synchronises: do this before capturing a step into a HIP graph, and
    whenever the first step's latency matters (the ops otherwise build the same things lazily on
    first use).  Returns (plan_r, plan_c)."""
    from . import _lib
    code = _lib.F32 if dtype == torch.float32 else _lib.F64
    plan_r = _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r, g.n_dst)
    plan_c = _lib.get_plan(g.col, g.ptr_c, g.eid_c, g.indices_c, g.n_src)
    plan_r.prepare(code, g.n_dst, h, d, 2 if fused else 0)      # the row-major side gathers the column-node table
    plan_c.prepare(code, g.n_src, h, d, 3 if fused else 0)
    return plan_r, plan_c


def release(g):
    """Drop this graph's plans (their device memory returns to torch's allocator)."""
    from . import _lib
    _lib.release_plans(g.row, g.col)


# ---- on-disk container (next-row N4: the reference caches its index arrays in `i.pt`,
# wrapper.py:114-116; here one file holds both chunked CSR orientations AND the derived plan state:
# info records, row-segment table, 32-bit index mirrors, every column-window structure built so far,
# block-dense cover) -----------------------------------------------------------------------------------
_GRAPH_FORMAT = 2
_INDEX_FIELDS = ("row", "ptr_r", "eid_r", "indices_r", "col", "ptr_c", "eid_c", "indices_c")


def save_graph(g, path, with_plans=True):
    """Write an AttnGraph (all index tensors, on CPU) with torch.save.  with_plans: if the graph is
    on a GPU and has plans (graphs.prepare or any op call), their exported state is stored too, so
    load_graph(..., device=gpu) re-creates them without analysing the graph again."""
    payload = {"format": _GRAPH_FORMAT, "plans": None}
    for k, v in g.__dict__.items():
        payload[k] = v.cpu() if isinstance(v, torch.Tensor) else v
    if with_plans and g.row.is_cuda:
        from . import _lib
        plans = {}
        for side, (row, ptr_, eid, idx) in (("r", (g.row, g.ptr_r, g.eid_r, g.indices_r)),
                                            ("c", (g.col, g.ptr_c, g.eid_c, g.indices_c))):
            for ip, ei, ix, _ver, p in row.__dict__.get("_graphop_plans", []):
                if ip is ptr_ and ei is eid and ix is idx:
                    plans[side] = p.export_state()
        payload["plans"] = plans or None
    torch.save(payload, path)


def load_graph(path, device="cpu"):
    """Read a container.  On a GPU device the stored plan state (if any) is imported: the first op
    call then neither validates the graph nor builds index mirrors or window structures."""
    payload = torch.load(path, map_location="cpu")
    fmt = payload.pop("format", None)
    if fmt not in (1, _GRAPH_FORMAT):
        raise RuntimeError("load_graph: %s is not a graph container of format <= %d" % (path, _GRAPH_FORMAT))
    plans = payload.pop("plans", None)
    g = AttnGraph(**payload)
    if str(device) == "cpu":
        return g
    g = g.to(device)
    if plans:
        from . import _lib
        if "r" in plans:
            _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r, g.n_dst, state=plans["r"])
        if "c" in plans:
            _lib.get_plan(g.col, g.ptr_c, g.eid_c, g.indices_c, g.n_src, state=plans["c"])
    return g


# ---- dlpack interchange (the reference's README TODO "Switch backend to dlpack", README.md:5-7):
# the eight index arrays of a graph as DLPack capsules / from any __dlpack__ producer -----------------
def to_dlpack(g):
    """{name: DLPack capsule} of the eight chunked-CSR arrays, in Function.apply order (zero-copy)."""
    from torch.utils import dlpack
    return {k: dlpack.to_dlpack(getattr(g, k)) for k in _INDEX_FIELDS}


def from_dlpack(arrays, n_src, n_dst=None, chunk_size=32):
    """AttnGraph over arrays exported by another framework: `arrays` maps the eight names (row, ptr_r,
    eid_r, indices_r, col, ptr_c, eid_c, indices_c) to DLPack capsules or objects with __dlpack__
    (zero-copy; int64 as in the reference API).  indptr / src / dst are re-derived."""
    n_dst = n_src if n_dst is None else n_dst
    t = {k: torch.from_dlpack(arrays[k]) for k in _INDEX_FIELDS}
    for k, v in t.items():
        if v.dtype != torch.int64:
            raise RuntimeError("from_dlpack: %s must be int64 (got %s)" % (k, v.dtype))
    dev = t["row"].device
    E = int(t["eid_r"].numel())

    def indptr_of(row, ptr, n):
        # rows' slot ranges from the chunk list (chunks of a row are adjacent, part_csr.py:18-21)
        ip = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if row.numel():
            lens = ptr[1:] - ptr[:-1]
            ip[1:] = torch.cumsum(torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row, lens), 0)
        return ip

    indptr_r = indptr_of(t["row"], t["ptr_r"], n_src)
    indptr_c = indptr_of(t["col"], t["ptr_c"], n_dst)
    src = torch.repeat_interleave(torch.arange(n_src, device=dev), indptr_r[1:] - indptr_r[:-1])
    return AttnGraph(n_src, n_dst, E, src, t["indices_r"], indptr_r, indptr_c, t["row"], t["ptr_r"], t["eid_r"],
                     t["indices_r"], t["col"], t["ptr_c"], t["eid_c"], t["indices_c"], chunk_size)
