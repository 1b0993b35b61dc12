"""Monte-Carlo harness for the QLDPC path.

`sim_ber(..., qldpc=True)` / `count_block_errors` / `PlotBER.simulate` keep the call contracts of
/root/reference sionna/utils/misc.py:403-768, sionna/utils/metrics.py:194-223 and
sionna/utils/plotting.py:312-447 for the quantum branch only (flagged / logical error counting of
misc.py:647-654).  With ``dist=True`` the per-batch counters are summed over all ranks of the default
torch.distributed process group (RCCL over xGMI on MI355X, gloo in the CPU tests).
"""
import time

import numpy as np
import torch


def count_block_errors(b, b_hat):
    """Number of rows (last axis = one block) in which b and b_hat differ (metrics.py:194-223)."""
    return (b != b_hat).any(dim=-1).to(torch.int64).sum()


def allreduce_counts(counts, force=False):
    """Sum a small integer tensor over all ranks (no-op without an initialised process group, and — unless ``force`` —
    with a single rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size() > 1):
        if counts.is_cuda and dist.get_backend() != "nccl":
            # host-memory backends (gloo: CPU tests, or several ranks sharing one GPU) reduce through a CPU copy
            host = counts.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            counts.copy_(host)
        else:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def _require_device_u8(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous uint8 [B, n] tensor on a HIP device (the bit-pack kernel has no CPU path)")


def pack_decisions(x_hat, z_hat):
    """[B, ceil(2n/8)] uint8 on the same device: row b = the 2n bits [x_hat[b] | z_hat[b]], numpy.packbits order
    (fgnn_pack_decisions).  2n bits per codeword is what a rank ships when decisions are collected on every rank
    (318 B per [[1270,28]] codeword)."""
    import ctypes as C
    from . import _lib
    _require_device_u8(x_hat, "x_hat")
    _require_device_u8(z_hat, "z_hat")
    if x_hat.shape != z_hat.shape or x_hat.device != z_hat.device:
        raise ValueError("x_hat and z_hat must have the same shape and device")
    B, n = int(x_hat.shape[0]), int(x_hat.shape[1])
    packed = torch.empty((B, (2 * n + 7) // 8), dtype=torch.uint8, device=x_hat.device)
    with torch.cuda.device(x_hat.device):
        _lib.check(_lib.lib().fgnn_pack_decisions(C.c_void_p(x_hat.data_ptr()), C.c_void_p(z_hat.data_ptr()), B, n,
                                                  C.c_void_p(packed.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream(x_hat.device).cuda_stream)))
    return packed


def unpack_decisions(packed, n):
    """Inverse of `pack_decisions`: (x_hat, z_hat) uint8 [B, n] on the device of ``packed`` (fgnn_unpack_decisions)."""
    import ctypes as C
    from . import _lib
    _require_device_u8(packed, "packed")
    B, n = int(packed.shape[0]), int(n)
    if packed.shape[1] != (2 * n + 7) // 8:
        raise ValueError(f"packed must have {(2 * n + 7) // 8} bytes per codeword for n = {n}")
    x_hat = torch.empty((B, n), dtype=torch.uint8, device=packed.device)
    z_hat = torch.empty((B, n), dtype=torch.uint8, device=packed.device)
    with torch.cuda.device(packed.device):
        _lib.check(_lib.lib().fgnn_unpack_decisions(C.c_void_p(packed.data_ptr()), B, n, C.c_void_p(x_hat.data_ptr()),
                                                    C.c_void_p(z_hat.data_ptr()),
                                                    C.c_void_p(torch.cuda.current_stream(packed.device).cuda_stream)))
    return x_hat, z_hat


def gather_packed(packed):
    """All-gather of per-rank rows ``packed [B_r, nb]`` (uint8) in rank order onto every rank: ``[sum_r B_r, nb]``.  One
    `all_gather_into_tensor` (RCCL over xGMI with the nccl backend); ranks may own different numbers of rows (the shard sizes
    of `shard_range` differ by at most one) — then the rows are padded to the largest shard for the collective and the padding is
    dropped afterwards.  Without a process group (or with world size 1 and no group) the input is returned unchanged."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return packed
    world = dist.get_world_size()
    via_host = packed.is_cuda and dist.get_backend() != "nccl"  # gloo: several ranks sharing one GPU, or the CPU tests
    buf = packed.cpu() if via_host else packed
    sizes = torch.zeros(world, dtype=torch.int64, device=buf.device)
    mine = torch.tensor([buf.shape[0]], dtype=torch.int64, device=buf.device)
    dist.all_gather_into_tensor(sizes, mine)
    sizes = [int(v) for v in sizes.cpu()]
    most, nb = max(sizes), int(buf.shape[1])
    if buf.shape[0] < most:
        buf = torch.cat([buf, torch.zeros((most - buf.shape[0], nb), dtype=buf.dtype, device=buf.device)])
    out = torch.empty((world * most, nb), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf.contiguous())
    if any(sz != most for sz in sizes):
        out = torch.cat([out[r * most:r * most + sizes[r]] for r in range(world)])
    return out.to(packed.device) if via_host else out


def gather_decisions(x_hat, z_hat):
    """Decisions of ALL ranks on every rank, in rank (= global sample) order: ``(x_hat_all, z_hat_all)`` uint8
    ``[sum_r B_r, n]``.  The ranks exchange 2n bits per codeword: bit-pack kernel -> one all-gather -> unpack kernel."""
    n = int(x_hat.shape[1])
    return unpack_decisions(gather_packed(pack_decisions(x_hat, z_hat)), n)


def broadcast_weights(system, src=0, device=None):
    """One set of weights on every rank: rank ``src``'s ``system.get_weights()`` (a list of float32 arrays, the Keras order of
    `Feedback_GNN` / `GNN_BP4`) is sent to all ranks with ONE `torch.distributed.broadcast` of the concatenated parameters (3 923
    floats = 15.7 KB for the shipped feedback GNN; RCCL over xGMI with the nccl backend, through host memory with gloo) and installed
    with ``system.set_weights``.  The reference evaluates one process per GPU, each loading the same pickle (n882.py:9-25, :52); here a
    job whose rank 0 alone holds the weights — just trained (`training.py`), or read from a path only it can see — ships them instead.
    Every rank must have constructed ``system`` with the same architecture: the array shapes come from its own ``get_weights()`` and
    the total size is checked across ranks before anything is overwritten.  No-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return system
    mine = [np.ascontiguousarray(w, dtype=np.float32) for w in system.get_weights()]
    shapes = [w.shape for w in mine]
    total = int(sum(w.size for w in mine))
    nccl = dist.get_backend() == "nccl"
    dev = (torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)) if nccl else torch.device("cpu")
    # every rank's parameter count against the source's, before a single weight is touched
    counts = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([total], dtype=torch.int64, device=dev))
    counts = [int(v) for v in counts.cpu()]
    if any(c != counts[src] for c in counts):
        raise ValueError(f"broadcast_weights: the ranks hold different architectures (parameter counts {counts})")
    flat = torch.from_numpy(np.concatenate([w.ravel() for w in mine]) if mine else np.zeros(0, np.float32)).to(dev)
    dist.broadcast(flat, src=src)
    flat = flat.cpu().numpy()
    out, at = [], 0
    for shp in shapes:
        k = int(np.prod(shp))
        out.append(flat[at:at + k].reshape(shp).copy())
        at += k
    system.set_weights(out)
    return system


def shard_range(total, rank, world_size):
    """Contiguous slice [lo, hi) of ``total`` samples owned by ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sim_ber(mc_fun, ebno_dbs, batch_size, max_mc_iter, soft_estimates=False, num_target_bit_errors=None,
            num_target_block_errors=None, early_stop=True, graph_mode=None, verbose=True,
            forward_keyboard_interrupt=True, qldpc=True, dist=False, dtype=None, device_counters=True, max_deferred=64,
            fuse_samples=65536):
    """Simulate until the target number of errors is reached; returns (flagged_rate, bler) per point.

    Only the ``qldpc=True`` branch of the reference exists here: ``mc_fun(batch_size=, ebno_db=)`` returns
    ``(s_hat, ls_hat)`` and a block counts as flagged / as a block error iff its row is non-zero (misc.py:636-738).

    Device-counter path (``device_counters=True`` and ``mc_fun`` offers ``mc_step(batch_size, p, counts)`` and
    ``rewind(batches, batch_size)``, as `Sandwich_BP_GNN_Evaluation_Model` does): the three counters stay on the device,
    every batch leaves a snapshot of them in a device ring, and the host reads the ring back (one copy, one all-reduce with
    ``dist=True``) only when the target could have been reached — after the first batch, then after as many batches as the
    measured error rate predicts, at most ``max_deferred``.  The stopping rule is then applied to the snapshots IN ORDER, so
    the point ends after exactly the batch at which the per-batch loop of the reference would have ended; batches run beyond
    it are discarded and the model's sample stream is rewound, so every counter of every point equals the per-batch path's.
    """
    if not qldpc:
        raise NotImplementedError("only the qldpc=True branch of sim_ber is part of this package")
    ps = [float(p) for p in np.atleast_1d(np.asarray(ebno_dbs, dtype=np.float64))]
    n_pts = len(ps)
    flag_errors = np.zeros(n_pts, np.int64)
    block_errors = np.zeros(n_pts, np.int64)
    nb_blocks = np.zeros(n_pts, np.int64)
    runtime = np.zeros(n_pts)
    status = np.zeros(n_pts, int)
    names = ["not simulated", "reached max iter       ", "no errors - early stop", "reached target bit errors",
             "reached target block errors"]
    header = ["p", "Flagged", "BLER", "flag errors", "block errors", "num blocks", "runtime [s]", "status"]
    fast = bool(device_counters) and hasattr(mc_fun, "mc_step") and hasattr(mc_fun, "rewind")
    fuse = int(fuse_samples) // max(int(batch_size), 1) if (fast and hasattr(mc_fun, "mc_steps")) else 0
    max_it = int(max_mc_iter)
    # A model that issues `mc_step` on side streams (streams > 1) adds to the shared counters from those streams: every snapshot below
    # is ordered behind them first, or it could miss a batch or tear a counter triple.  (A device-side wait; the per-batch snapshots
    # the stopping rule needs serialise the batches anyway — the fused `mc_steps` launch is what keeps the chip full here.)
    join = getattr(mc_fun, "join", None) if fast else None

    def row(i, st):
        fl = flag_errors[i] / max(nb_blocks[i], 1)
        bl = block_errors[i] / max(nb_blocks[i], 1)
        return (f"{ps[i]:9.4g} | {fl:10.4e} | {bl:10.4e} | {flag_errors[i]:11d} | {block_errors[i]:12d} | "
                f"{nb_blocks[i]:11d} | {runtime[i]:11.1f} |{st}")

    def stop_status(c, it):
        """Status after a batch whose cumulative counters are c (the reference's checks, misc.py:700-738), 0 = go on."""
        if num_target_bit_errors is not None and c[0] >= num_target_bit_errors:
            return 3
        if num_target_block_errors is not None and c[1] >= num_target_block_errors:
            return 4
        return 1 if it == max_it - 1 else 0

    def run_point_per_batch(i, t0):
        for it in range(max_it):
            s_hat, l_hat = mc_fun(batch_size=batch_size, ebno_db=ps[i])[:2]
            c = torch.stack([count_block_errors(torch.zeros_like(s_hat), s_hat),
                             count_block_errors(torch.zeros_like(l_hat), l_hat),
                             torch.tensor(s_hat.shape[0], dtype=torch.int64, device=s_hat.device)])
            if dist:
                c = allreduce_counts(c)
            c = c.cpu().numpy()
            flag_errors[i] += c[0]
            block_errors[i] += c[1]
            nb_blocks[i] += c[2]
            runtime[i] = time.perf_counter() - t0
            st = stop_status((flag_errors[i], block_errors[i]), it)
            if st:
                status[i] = st
                break

    def run_point_device(i, t0):
        counts = None
        ring = None
        it = 0       # batches issued
        done = 0     # batches whose snapshot the host has seen
        k = 1
        while it < max_it:
            k = max(1, min(k, max_it - it, int(max_deferred)))
            j = 0
            while j < k:
                if counts is None:
                    probe = mc_fun.mc_step(batch_size, ps[i], None)  # allocates the device counters on the model's device
                    counts = probe
                    ring = torch.zeros((int(max_deferred), 3), dtype=torch.int64, device=counts.device)
                    if join is not None:
                        join()
                    ring[0].copy_(counts)
                    j += 1
                elif fuse > 1 and k - j > 1:
                    kk = min(fuse, k - j)
                    mc_fun.mc_steps(batch_size, ps[i], kk, counts, ring[j:j + kk])
                    j += kk
                else:
                    mc_fun.mc_step(batch_size, ps[i], counts)
                    if join is not None:
                        join()
                    ring[j].copy_(counts)
                    j += 1
            it += k
            if join is not None:
                join()
            snap = ring[:k].clone()
            if dist:
                snap = allreduce_counts(snap)
            snap = snap.cpu().numpy()  # the only host synchronisation of these k batches
            ended = False
            for j in range(k):
                st = stop_status(snap[j], done + j)
                if st:
                    flag_errors[i], block_errors[i], nb_blocks[i] = snap[j]
                    status[i] = st
                    if k - 1 - j:
                        mc_fun.rewind(k - 1 - j, batch_size)  # batches issued beyond the one the point ends with
                    ended = True
                    break
            runtime[i] = time.perf_counter() - t0
            if ended:
                return
            flag_errors[i], block_errors[i], nb_blocks[i] = snap[k - 1]
            done += k
            # next read-back when the nearer target is expected to be reached, from the rate seen so far
            want = []
            if num_target_bit_errors is not None:
                want.append((num_target_bit_errors - flag_errors[i]) / max(flag_errors[i] / done, 1e-9))
            if num_target_block_errors is not None:
                want.append((num_target_block_errors - block_errors[i]) / max(block_errors[i] / done, 1e-9))
            k = int(min(want)) if want else int(max_deferred)
            k = max(1, min(k, 4 * done))  # never commit to more than 4x what has been seen

    try:
        for i in range(n_pts):
            t0 = time.perf_counter()
            if verbose and i == 0:
                print(" | ".join(f"{h:>11s}" for h in header))
                print("-" * 135)
            (run_point_device if fast else run_point_per_batch)(i, t0)
            if verbose:
                print(row(i, names[status[i]]))
            if early_stop and block_errors[i] == 0:
                status[i] = 2
                if verbose:
                    print(f"\nSimulation stopped as no error occurred @ p = {ps[i]:.4g}.\n")
                break
    except KeyboardInterrupt:
        if forward_keyboard_interrupt:
            raise
        print("\nSimulation stopped by the user")
    with np.errstate(invalid="ignore", divide="ignore"):
        flagged = np.where(nb_blocks > 0, flag_errors / np.maximum(nb_blocks, 1), 0.0)
        bler = np.where(nb_blocks > 0, block_errors / np.maximum(nb_blocks, 1), 0.0)
    sim_ber.last = dict(p=ps, flag_errors=flag_errors, block_errors=block_errors, num_blocks=nb_blocks, runtime=runtime,
                        status=status, device_counters=fast)
    return flagged, bler


class PlotBER:
    """Result store with the `simulate` entry point of sionna/utils/plotting.py:312-447 (qldpc branch);
    keeps `_snrs`, `_bers`, `_legends` like the reference (two entries per run: flagged, then BLER)."""

    def __init__(self, title="Logical error rate"):
        self._title = title
        self._bers, self._snrs, self._legends, self._is_bler = [], [], [], []

    def simulate(self, mc_fun, ebno_dbs, batch_size, max_mc_iter, legend="", add_ber=True, add_bler=False,
                 soft_estimates=False, num_target_bit_errors=None, num_target_block_errors=None, early_stop=True,
                 graph_mode=None, add_results=True, forward_keyboard_interrupt=True, show_fig=False, verbose=True,
                 qldpc=True, dist=False, device_counters=True, fuse_samples=65536):
        flagged, bler = sim_ber(mc_fun, ebno_dbs, batch_size, max_mc_iter, soft_estimates=soft_estimates,
                                num_target_bit_errors=num_target_bit_errors,
                                num_target_block_errors=num_target_block_errors, early_stop=early_stop, verbose=verbose,
                                forward_keyboard_interrupt=forward_keyboard_interrupt, qldpc=qldpc, dist=dist,
                                device_counters=device_counters, fuse_samples=fuse_samples)
        if add_results:
            ps = np.atleast_1d(np.asarray(ebno_dbs, dtype=np.float64))
            if add_ber:
                self._bers.append(flagged); self._snrs.append(ps); self._legends.append(legend); self._is_bler.append(False)
            if add_bler:
                self._bers.append(bler); self._snrs.append(ps); self._legends.append(legend + " (BLER)"); self._is_bler.append(True)
        return flagged, bler

    # ---- the rest of the reference class' surface (plotting.py:207-310, :449-504): result bookkeeping and the figure ----
    title = property(lambda self: self._title)

    @title.setter
    def title(self, value):
        if not isinstance(value, str):
            raise AssertionError("title must be string")
        self._title = value

    ber = property(lambda self: self._bers)
    snr = property(lambda self: self._snrs)
    legend = property(lambda self: self._legends)
    is_bler = property(lambda self: self._is_bler)

    def add(self, ebno_db, ber, is_bler=False, legend=""):
        """Store an externally computed curve (x values, error rates)."""
        if len(np.atleast_1d(ebno_db)) != len(np.atleast_1d(ber)):
            raise AssertionError("ebno_db and ber must have same number of elements.")
        if not isinstance(legend, str) or not isinstance(is_bler, bool):
            raise AssertionError("legend must be str and is_bler must be bool.")
        self._snrs.append(np.asarray(ebno_db, dtype=np.float64))
        self._bers.append(np.asarray(ber, dtype=np.float64))
        self._legends.append(legend)
        self._is_bler.append(is_bler)

    def reset(self):
        self._bers, self._snrs, self._legends, self._is_bler = [], [], [], []

    def remove(self, idx=-1):
        if not isinstance(idx, int):
            raise AssertionError("idx must be int.")
        for store in (self._bers, self._snrs, self._legends, self._is_bler):
            del store[idx]

    def __call__(self, snr_db=(), ber=(), legend=(), is_bler=(), show_ber=True, show_bler=True, xlim=None, ylim=None,
                 save_fig=False, path=""):
        """Semilog-y figure of the stored curves plus the ones passed in (x axis: physical error rate p, which the reference
        carries in its ``ebno_db`` argument).  Returns the matplotlib figure; ``save_fig`` writes it to ``path``."""
        import matplotlib
        if save_fig or not hasattr(matplotlib, "pyplot"):
            matplotlib.use(matplotlib.get_backend() if "inline" in matplotlib.get_backend().lower() else "Agg", force=False)
        import matplotlib.pyplot as plt

        def as_list(x, n=None):
            if isinstance(x, np.ndarray) or (len(x) and np.isscalar(x[0])):
                x = [x]
            x = list(x)
            return x * n if n and len(x) == 1 and n > 1 else x

        bers = self._bers + as_list(ber)
        extra = len(bers) - len(self._bers)
        snrs = self._snrs + as_list(snr_db, extra)
        legends = self._legends + ([legend] if isinstance(legend, str) else list(legend))
        flags = self._is_bler + ([is_bler] if isinstance(is_bler, bool) else list(is_bler))
        legends += [""] * (len(bers) - len(legends))
        flags += [False] * (len(bers) - len(flags))
        keep = [i for i, f in enumerate(flags) if (f and show_bler) or (not f and show_ber)]
        ylabel = "BLER" if keep and all(flags[i] for i in keep) else ("BER" if not any(flags[i] for i in keep) else "BER / BLER")
        fig, ax = plt.subplots(figsize=(10, 6))
        for i in keep:
            ax.semilogy(np.asarray(snrs[i], dtype=np.float64), np.asarray(bers[i], dtype=np.float64), "--" if flags[i] else "-",
                        marker="o", markersize=4, label=legends[i])
        ax.set_title(self._title)
        ax.set_xlabel("physical error rate p")
        ax.set_ylabel(ylabel)
        ax.grid(which="both", alpha=0.4)
        if xlim is not None:
            ax.set_xlim(xlim)
        if ylim is not None:
            ax.set_ylim(ylim)
        if any(legends[i] for i in keep):
            ax.legend()
        if save_fig:
            fig.savefig(path)
        return fig

