"""OSD-0 post-processing and the BP4+OSD evaluation model on MI355X — SURVEY.md §8(f) rank 2.

Drop-ins for `OSD0_Decoder` and `BP4_OSD_Model` of /root/reference sionna/fec/ldpc/bp_osd.py:8-191.
"""
import numpy as np
import torch

from ._lib import ROWS_LX, ROWS_LZ
from .feedback_gnn import Pauli


class OSD0_Decoder:
    """Order-0 ordered-statistics decoder (bp_osd.py:8-77).  Inside `BP4_OSD_Model` / `BP2_OSD_Model` the row basis lives in the
    model's device graph (`fgnn_graph_set_basis`) and only the BP failures are re-solved; the reference's standalone
    ``decoder(llr, pcm, s, bs)`` is served by `__call__` below through the same HIP kernel (`fgnn_osd0`)."""

    MAX_CACHED = 4  # device graphs kept (a script alternates the hx and the hz basis, bp_osd.py:147-157: two)

    def __init__(self, n, device=None):
        self.n = int(n)
        self._device = device
        self._graphs = {}  # basis bytes -> device graph with the basis installed (a caller reuses its bases for every batch)
        self._seen = {}    # identity of a pcm tensor already resolved (storage pointer, shape, strides, version) -> its graph

    def _graph_of(self, basis):
        from .decoding import _binary_graph
        key = (basis.shape, basis.tobytes())
        g = self._graphs.pop(key, None)
        if g is None:
            g = _binary_graph(basis, None, self._device)
            g.set_basis(0, np.arange(basis.shape[0], dtype=np.int32))
        self._graphs[key] = g  # most recently used last
        while len(self._graphs) > self.MAX_CACHED:
            old = next(iter(self._graphs))
            self._seen = {k: v for k, v in self._seen.items() if v is not self._graphs[old]}
            del self._graphs[old]
        return g

    def __call__(self, llr, pcm, s, bs=None):
        """The reference's standalone call (bp_osd.py:47-77): ``llr [bs, n]`` binary reliabilities (sorted ascending: the least
        reliable "no error" positions become the pivots), ``pcm [bs, rank, n]`` the FULL-RANK row basis tiled over the batch (the
        reference's models tile one matrix, :147-150; this implementation requires that — or takes a plain ``[rank, n]`` matrix),
        ``s [rank, bs]`` the syndrome of those rows → ``e_hat [bs, n]`` bool with ``pcm e_hat = s`` on the most reliable basis.
        Ties in the sort keep qubit order (tf.argsort leaves them unspecified).

        Host-side cost: the first call with a given ``pcm`` tensor checks that it is one matrix tiled over the batch (skipped for an
        expanded, stride-0 batch dimension), copies it to the host once and builds (or finds) its device graph; later calls with the
        SAME tensor (same storage, shape, strides and version counter) go straight to the kernel — no device synchronisation, no copy,
        no hash."""
        pcm_t = torch.as_tensor(pcm)
        ident = (pcm_t.untyped_storage().data_ptr(), pcm_t.storage_offset(), tuple(pcm_t.shape), tuple(pcm_t.stride()), pcm_t._version,
                 pcm_t.dtype, str(pcm_t.device))
        # only a torch tensor has a version counter that sees in-place writes; anything else (a NumPy array ...) is resolved by content
        is_tensor = isinstance(pcm, torch.Tensor)
        g = self._seen.get(ident) if is_tensor else None
        if g is None:
            if pcm_t.dim() == 3:
                tiled = pcm_t.shape[0] <= 1 or pcm_t.stride(0) == 0 or bool((pcm_t == pcm_t[:1]).all())
                if not tiled:
                    raise NotImplementedError("OSD0_Decoder: one row basis per call (the reference tiles the same matrix over the batch)")
                pcm_t = pcm_t[0]
            basis = np.ascontiguousarray(pcm_t.cpu().numpy() != 0, dtype=np.uint8)
            if basis.shape[1] != self.n:
                raise ValueError("pcm must have n columns")
            g = self._graph_of(basis)
            if len(self._seen) > 64:
                self._seen.clear()
            if is_tensor:
                self._seen[ident] = g
        rank = g.m_x
        llr = torch.as_tensor(llr, device=g.device).to(torch.float32).contiguous()
        B = int(llr.shape[0])
        if bs is not None and int(bs) != B:
            raise ValueError("bs must equal the leading dimension of llr")
        synd = (torch.as_tensor(s, device=g.device).to(torch.int64) & 1).to(torch.uint8).t().contiguous()
        if tuple(synd.shape) != (B, rank):
            raise ValueError(f"s must have shape [{rank}, {B}]")
        e_hat = torch.zeros((B, self.n), dtype=torch.uint8, device=g.device)
        g.osd0(0, synd, e_hat, llr_bin=llr)
        return e_hat.bool()

    call = __call__


class BP4_OSD_Model:
    """``BP4_OSD_Model(code, bp4_decoder, osd_decoder)``; ``model(batch_size, p)`` → ``(zeros_like(ls_hat), ls_hat)`` with
    ``ls_hat[bs, rows(lz)+rows(lx)] = [lz·x_diff ; lx·z_diff]`` (bp_osd.py:159-191).  BP4 runs on every sample with
    ``llr = log(3(1-p)/p)`` (:106); the samples whose estimate misses the syndrome get both halves re-solved by OSD-0 from
    the binary reliabilities of their BP marginals (:117-131, :138-157)."""

    def __init__(self, code, bp4_decoder, osd_decoder, *, seed=0x5EED, rank=0, world_size=1):
        self.code, self.bp4_decoder, self.osd_decoder = code, bp4_decoder, osd_decoder
        self.graph = bp4_decoder.graph
        self.graph.set_basis(0, code.pivot_hx)
        self.graph.set_basis(1, code.pivot_hz)
        self.channel = Pauli(self.graph, seed=seed)
        self.rank, self.world_size, self._next = int(rank), int(world_size), 0
        self.last_num_osd = 0

    def decode(self, batch_size, p):
        B, g, d = int(batch_size), self.graph, self.bp4_decoder
        first = self._next + self.rank * B
        self._next += self.world_size * B
        ex, ez = self.channel(B, float(p), first)
        sx, sz = g.syndrome(ex, ez)
        pf = np.float32(p)
        L = float(np.log(np.float32(3.0) * (np.float32(1.0) - pf) / pf, dtype=np.float32))
        out = g.bp4_decode(sx, sz, d.num_iter, d.cn_type, d.normalization_factor, llr_const=L, want_logits=False)
        x_hat, z_hat = out["x_hat"], out["z_hat"]
        _, _, flags = g.residual(ex, ez, x_hat, z_hat, want_arrays=False)  # bit 0 = syndrome missed = `err` (:117-120)
        index, nact = g.compact(flags, 1)
        self.last_num_osd = nact
        if nact:
            g.osd0(0, sx, z_hat, marg=out["llr"], index=index, nact=nact)  # z_hat_osd from hx, osd_llrz (:155)
            g.osd0(1, sz, x_hat, marg=out["llr"], index=index, nact=nact)  # x_hat_osd from hz, osd_llrx (:156)
        return dict(noise_x=ex, noise_z=ez, x_hat=x_hat, z_hat=z_hat)

    def __call__(self, batch_size, ebno_db=None, **kw):
        o = self.decode(batch_size, kw.get("p", ebno_db))
        ls_hat, _ = self.graph.residual_rows(ROWS_LZ, ROWS_LX, o["noise_x"], o["noise_z"], o["x_hat"], o["z_hat"])
        return torch.zeros_like(ls_hat), ls_hat

    call = __call__


class BP2_OSD_Model:
    """``BP2_OSD_Model(pcm, pcm_basis, pivot_pcm, logical_pcm, bp2_decoder, osd_decoder)``; ``model(batch_size, p)`` →
    ``(zeros_like(ls_hat), ls_hat[bs, rows(logical_pcm)])`` (bp_osd.py:194-273): BSC(p) noise, binary syndrome BP with soft
    output, OSD-0 on the samples whose estimate misses the syndrome, ``ls_hat = logical_pcm·(noise xor estimate)``."""

    def __init__(self, pcm, pcm_basis, pivot_pcm, logical_pcm, bp2_decoder, osd_decoder, *, seed=0x5EED, rank=0, world_size=1):
        from .decoding import _binary_graph
        self.pcm, self.logical_pcm, self.bp2_decoder, self.osd_decoder = pcm, logical_pcm, bp2_decoder, osd_decoder
        self.graph = _binary_graph(pcm, logical_pcm, bp2_decoder.graph.device)
        self.graph.set_basis(0, pivot_pcm)
        self.seed, self.rank, self.world_size, self._next = int(seed), int(rank), int(world_size), 0
        self.last_num_osd = 0

    def __call__(self, batch_size, ebno_db=None, **kw):
        p = float(kw.get("p", ebno_db))
        B, g, d = int(batch_size), self.graph, self.bp2_decoder
        first = self._next + self.rank * B
        self._next += self.world_size * B
        pf = np.float32(p)
        llr_const = float(-np.log((np.float32(1.0) - pf) / pf, dtype=np.float32))  # (:216)
        noise = g.bsc_noise(self.seed, p, first, B)
        zeros = torch.zeros_like(noise)
        synd, _ = g.syndrome(zeros, noise)
        soft, noise_hat = g.bp2_decode(synd, d._num_iter, d._cn_type, d._normalization_factor, llr_const=llr_const, B=B)
        _, _, flags = g.residual(noise, zeros, noise_hat, zeros, want_arrays=False)
        index, nact = g.compact(flags, 1)
        self.last_num_osd = nact
        if nact:
            g.osd0(0, synd, noise_hat, llr_bin=(-soft).contiguous(), index=index, nact=nact)  # llr_hat = -decoder output (:225)
        _, ls_hat, _ = g.residual(noise, zeros, noise_hat, zeros, want_arrays=True)
        ls_hat = ls_hat[:, :g.rows_hxp].contiguous()
        return torch.zeros_like(ls_hat), ls_hat

    call = __call__
