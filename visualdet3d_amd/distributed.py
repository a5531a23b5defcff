"""Multi-GPU inference plumbing: frames are independent units, so a batch is split into contiguous
shards, one process per GPU (weights replicated), no data-path collective.  The only communication is the final gather
of the fixed-size padded detection tensors (a few tens of KB per rank: one direct all_gather over RCCL/xGMI -- latency
bound, nothing to tune).  Backend-agnostic: the same code runs over ``gloo`` on CPU tensors in the tests."""
import os

import torch
import torch.distributed as dist


# ---- host-side placement of the ranks of one node ---------------------------------------------------------------------------------
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if part:
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(bdf, sysfs='/sys'):
    """NUMA node of the PCI device ``bdf`` ('0000:c1:00.0'), -1 when the platform reports none."""
    return int(open(os.path.join(sysfs, 'bus/pci/devices', bdf, 'numa_node')).read().strip())


def rank_cpu_sets(bdfs, sysfs='/sys', allowed=None):
    """CPU affinity of every local rank, rank r driving the GPU ``bdfs[r]``: the CPUs of the GPU's NUMA node (pinned-memory copies and the
    launch thread stay local to the GPU's root complex), DIVIDED among the ranks whose GPUs share that node -- 8 ranks on a two-socket host are
    4 + 4, each with its own quarter of a socket, so that eight concurrent host-feed producers never compete for a core.  ``allowed``: the CPUs
    this process may use at all (cgroup / taskset); a rank whose slice would be empty keeps the node's whole allowed set; a GPU without a NUMA node
    -> None (not pinned)."""
    nodes = [gpu_numa_node(b, sysfs) for b in bdfs]
    out = []
    for r, node in enumerate(nodes):
        if node < 0:
            out.append(None)
            continue
        cpus = _parse_cpulist(open(os.path.join(sysfs, 'devices/system/node/node%d/cpulist' % node)).read())
        if allowed is not None:
            cpus &= set(allowed)
        peers = [i for i, n in enumerate(nodes) if n == node]
        order = sorted(cpus)
        per = len(order) // len(peers)
        slot = peers.index(r)
        mine = set(order[slot * per:(slot + 1) * per]) if per >= 1 else set()
        out.append(mine or cpus or None)
    return out


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) shard of n frames for ``rank`` (first n % world ranks get one extra frame)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_detections(scores, boxes, labels, count, k):
    """padded (scores [B,K], boxes [B,K,11], labels [B,K], count [B]) -> one float tensor [B, k, 13] (rows past a frame's
    count zeroed BY SELECTION: the padding of the device buffers is uninitialised memory and may hold NaN / Inf, which a
    multiply by 0 would keep) + count [B]."""
    k = min(k, scores.shape[1])
    pack = torch.cat([scores[:, :k, None], boxes[:, :k], labels[:, :k, None].to(scores.dtype)], dim=2).contiguous()
    valid = torch.arange(k, device=count.device)[None, :] < count[:, None]
    return pack.masked_fill(~valid[:, :, None], 0), torch.clamp(count, max=k)


def gather_detections(pack, count, group=None):
    """all_gather of equally-shaped per-rank packs; returns ([world*B, k, 13], [world*B]) in rank order."""
    world = dist.get_world_size(group)
    packs = [torch.empty_like(pack) for _ in range(world)]
    counts = [torch.empty_like(count) for _ in range(world)]
    dist.all_gather(packs, pack, group=group)
    dist.all_gather(counts, count, group=group)
    return torch.cat(packs, dim=0), torch.cat(counts, dim=0)


def _check_counts(counts):
    """head_postprocess marks a frame whose candidate / detection list overflowed with a NEGATIVE count; such a frame's rows
    are undefined.  Same error as AnchorBasedDetection3DHead.unpad."""
    bad = [i for i, n in enumerate(counts) if n < 0]
    if bad:
        raise RuntimeError('frame(s) %s: more candidates than max_candidates (or detections than max_det); raise '
                           'AnchorBasedDetection3DHead.max_candidates' % bad)


def unpack_detections(pack, count):
    """-> list of per-frame (scores[N], boxes[N,11], labels[N] int64).  Raises on overflow-marked (negative) counts."""
    out = []
    _check_counts(count.tolist())
    for b, n in enumerate(count.tolist()):
        out.append((pack[b, :n, 0], pack[b, :n, 1:12], pack[b, :n, 12].long()))
    return out


class DetectionGather:
    """Per-step gather with ONE collective and no allocation: the per-rank pack is [B, k + 1, 13] floats, row k of every frame
    carrying the frame's detection count, all_gather'ed into a preallocated [world, B, k + 1, 13] buffer
    (``all_gather_into_tensor`` over RCCL; the list form of ``all_gather`` on backends without the tensor form, i.e. gloo on
    CPU tensors in the tests).  Which of the two is used is decided ONCE, from the backend, at construction: a collective that
    fails at run time propagates -- silently switching to a different collective on one rank would mismatch its peers."""

    def __init__(self, B, k, device, world=None, group=None):
        self.group = group
        self.world = world or dist.get_world_size(group)
        self.B, self.k = B, k
        self.pack = torch.zeros((B, k + 1, 13), dtype=torch.float32, device=device)
        self.out = torch.zeros((self.world, B, k + 1, 13), dtype=torch.float32, device=device)
        backend = str(dist.get_backend(group)) if dist.is_initialized() else 'none'
        # per-device backend strings ('cpu:gloo,cuda:nccl') count as RCCL for device buffers
        self.tensor_collective = ('nccl' in backend) and torch.device(device).type == 'cuda'

    def fill(self, scores, boxes, labels, count):
        """Pack one step's padded results into the send buffer (stream-ordered; no collective).  Device tensors: ONE
        ``vd3d_pack_detections`` launch (capturable in the step's hipGraph); CPU tensors (gloo tests): plain copies."""
        k, p = self.k, self.pack
        if scores.is_cuda:
            from . import hip_ops
            hip_ops.pack_detections(scores, boxes, labels, count, k, out=p)      # K < k allowed: rows K .. k-1 are zero
            return
        kk = min(k, scores.shape[1])
        p[:, :k].zero_()
        valid = torch.arange(kk)[None, :] < count[:, None].clamp(min=0)          # selection, not a multiply: padding may be NaN
        p[:, :kk, 0] = scores[:, :kk].masked_fill(~valid, 0)
        p[:, :kk, 1:12] = boxes[:, :kk].masked_fill(~valid[:, :, None], 0)
        p[:, :kk, 12] = labels[:, :kk].to(torch.float32).masked_fill(~valid, 0)
        p[:, k, 1:] = 0
        p[:, k, 0] = count.to(torch.float32)                 # negative (overflow marker) survives the round trip

    def gather(self):
        """The collective on the current stream; returns the [world, B, k + 1, 13] buffer."""
        p = self.pack
        if self.tensor_collective:
            dist.all_gather_into_tensor(self.out.view(-1), p.view(-1), group=self.group)
        else:
            dist.all_gather(list(self.out.unbind(0)), p, group=self.group)
        return self.out

    def __call__(self, scores, boxes, labels, count):
        self.fill(scores, boxes, labels, count)
        return self.gather()

    def counts(self):
        """[world, B] int32 detection counts (clamped to k) of the last gather; negative = overflow marker, kept negative."""
        return torch.clamp(self.out[:, :, self.k, 0].round().to(torch.int32), max=self.k)

    def detections(self):
        """-> ([world*B, k, 13], [world*B] counts) in rank order, like ``gather_detections``.  Raises on overflow-marked
        frames (one host sync for the counts)."""
        c = self.counts().reshape(-1)
        _check_counts(c.tolist())
        return self.out[:, :, :self.k].reshape(self.world * self.B, self.k, 13), c
