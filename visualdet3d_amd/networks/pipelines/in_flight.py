"""Steps in flight: a detector's forward as a replayable step, and several of them overlapped on one GPU.

Additive to the reference's test pipeline (`networks/pipelines/testers.py:10-42` runs one `module([...])` call at a time and waits for its
results).  A detector object owns its hipGraph, static buffers and post-processing scratch, so ONE object runs one step at a time; the chip,
however, is not full at a step's edges (select / NMS on a few CUs, the packed record's D2H, launch gaps, the 16 CUs a one-round strip GEMM
leaves idle).  `InFlight` holds k `CapturedStep`s -- k detector objects with the SAME weights and nothing else in common -- and replays them
round-robin on k streams: step i + 1 starts under the tail of step i.  Measured on MI355X (docs/rounds/r06.md section 7b): + 3 ... 6 % throughput
with k = 2 on BASELINE configs 2 and 5, nothing more with k = 3; the latency of a step doubles.  `bench.py` times exactly this loop.

    steps = [CapturedStep(model_a, inputs, B, own_inputs=True), CapturedStep(model_b, inputs, B, own_inputs=True)]
    pipe = InFlight(steps)
    t0 = pipe.submit(L0, R0, P2)          # enqueues copy-in, replay, record D2H on replica 0's stream; returns at once
    t1 = pipe.submit(L1, R1, P2)          # replica 1, overlapping
    dets0 = pipe.detections(t0)           # waits for step 0 only: per frame (scores [N], boxes [N, 11], labels [N])
"""
import time

import torch

from ... import hip_ops
from ... import distributed as vdist


class CapturedStep:
    """One detector object's step: `model.forward_device(*inputs)` (backbone ... decode + NMS, no host sync) + ONE launch that packs the padded
    results into a static `[B, k + 1, 13]` fp32 record (row k of a frame = its detection count), captured into a hipGraph after two eager warm-up
    passes.  `inputs` are the graph's static inputs: with `own_inputs=True` they are private clones that `set_inputs` refreshes (stream-ordered
    copies), otherwise the caller's tensors are read in place (a benchmark's resident batch).  `pre`: launches that open the captured step (e.g.
    the preprocessing of uploaded camera frames into the static inputs).  `side_pass`: one pass off the default stream before the capture, so that
    the model's own side streams exist when the capture begins."""

    def __init__(self, model, inputs, B, device=None, k=128, use_graph=True, pre=None, own_inputs=False, side_pass=True, sync_each_step=False):
        device = device if device is not None else inputs[0].device
        self.model, self.B, self.k, self.pre = model, B, k, pre
        self.inputs = tuple(t.clone() if (own_inputs and torch.is_tensor(t)) else t for t in inputs)
        self.own_inputs = own_inputs
        self.pack_static = torch.zeros((B, k + 1, 13), dtype=torch.float32, device=device)
        self.pinned = torch.empty((1, B, k + 1, 13), dtype=torch.float32).pin_memory()
        self.pinned_ring = [self.pinned, torch.empty((1, B, k + 1, 13), dtype=torch.float32).pin_memory()]
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.sync_each_step = sync_each_step
        self.graph = None

        def step_device():
            if self.pre is not None:
                self.pre()
            out = model.forward_device(*self.inputs)
            hip_ops.pack_detections(out[0], out[1], out[2], out[-1], k, out=self.pack_static)
            return out

        self.step_device = step_device
        with torch.no_grad():
            for _ in range(2):                       # packs weights, builds anchor tables, warms the allocator
                self.static_out = step_device()
            torch.cuda.synchronize()
            if use_graph:
                if side_pass:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        step_device()
                    torch.cuda.current_stream().wait_stream(side)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.static_out = step_device()

    def set_inputs(self, *inputs):
        """Refresh the static inputs (current stream; a None keeps the old value).  Needs `own_inputs=True`."""
        assert self.own_inputs, 'the static inputs belong to the caller (own_inputs=False)'
        assert len(inputs) == len(self.inputs)
        for dst, src in zip(self.inputs, inputs):
            if src is not None:
                dst.copy_(src, non_blocking=True)

    def forward_step(self):
        """Enqueue one step on the current stream; returns the (static) padded outputs of forward_device."""
        if self.graph is not None:
            self.graph.replay()
            return self.static_out
        with torch.no_grad():
            return self.step_device()

    def check(self, host_block):
        """host_block [ranks, B, k + 1, 13] (pinned): per-frame counts ride in row k.  -> counts [ranks, B]; a negative count is the overflow marker."""
        c = host_block[:, :, self.k, 0]
        assert float(c.min()) >= 0, 'candidate overflow in the head post-processing'
        return c.clone()

    def run(self, n, before_step=None):
        """n steps back to back on the current stream, each one's record copied to a pinned slot and checked on the host -- the host waits for step
        i - 1's record AFTER enqueuing step i (`sync_each_step`: before).  -> the counts of the last step."""
        counts = None
        main = torch.cuda.current_stream()
        for i in range(n):
            if before_step is not None:
                before_step(i)
            self.forward_step()
            s = 0 if self.sync_each_step else i & 1
            self.pinned_ring[s][0].copy_(self.pack_static, non_blocking=True)   # device -> host copy of the step's results
            if self.sync_each_step:
                main.synchronize()
                counts = self.check(self.pinned_ring[0])
                continue
            self.copied[s].record(main)
            if i >= 1:
                self.copied[s ^ 1].synchronize()                                # the one host sync per step: step i - 1's record is on the host
                counts = self.check(self.pinned_ring[s ^ 1])
        if n >= 1 and not self.sync_each_step:
            self.copied[(n - 1) & 1].synchronize()
            counts = self.check(self.pinned_ring[(n - 1) & 1])
        return counts

    def timed(self, steps, warmup, regions=3):
        """-> (median seconds of `regions` back-to-back timed regions of `steps` steps, every region's seconds, last counts)"""
        self.run(warmup)
        all_s, counts = [], None
        for _ in range(regions):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            counts = self.run(steps)
            torch.cuda.synchronize()
            all_s.append(time.perf_counter() - t0)
        return sorted(all_s)[len(all_s) // 2], all_s, counts


class InFlight:
    """k replicas (`CapturedStep`s of k detector objects with the same weights), each on its own stream; step i runs on replica i % k.  A replica's
    two pinned slots alternate, so a step's host record stays valid until 2 k further steps have been submitted."""

    def __init__(self, steps, streams=None):
        assert len(steps) >= 1
        self.steps = list(steps)
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream() for _ in steps]
        self.submitted = 0

    def __len__(self):
        return len(self.steps)

    def _slot(self, i):
        k = len(self.steps)
        return self.steps[i % k], self.streams[i % k], (i // k) & 1

    def submit(self, *inputs, before=None):
        """Enqueue one step (copy-in of `inputs` if given, replay, D2H of the record) on the next replica's stream; returns its ticket at once.
        `before(i, r)`: launches of the caller that must precede step i on replica r's stream (e.g. a refresh of what the step's `pre` reads)."""
        i = self.submitted
        st, s, slot = self._slot(i)
        if inputs:
            s.wait_stream(torch.cuda.current_stream())        # the caller's tensors were produced on the current stream
        with torch.cuda.stream(s):
            if before is not None:
                before(i, i % len(self.steps))
            if inputs:
                st.set_inputs(*inputs)
            st.forward_step()
            st.pinned_ring[slot][0].copy_(st.pack_static, non_blocking=True)
            st.copied[slot].record(s)
        self.submitted = i + 1
        return i

    def collect(self, ticket):
        """Wait for step `ticket`; -> its host record [1, B, k + 1, 13] (pinned; valid until 2 x replicas further submits)."""
        assert self.submitted - ticket <= 2 * len(self.steps), 'the record of this ticket has been overwritten'
        st, _, slot = self._slot(ticket)
        st.copied[slot].synchronize()
        return st.pinned_ring[slot]

    def counts(self, ticket):
        st = self._slot(ticket)[0]
        return st.check(self.collect(ticket))

    def detections(self, ticket):
        """-> per frame (scores [N], boxes [N, 11], labels [N] int64) of step `ticket` (raises on an overflow-marked frame)."""
        st = self._slot(ticket)[0]
        rec = self.collect(ticket)[0]
        count = torch.clamp(rec[:, st.k, 0].round().to(torch.int32), max=st.k)
        return vdist.unpack_detections(rec[:, :st.k], count)

    def run(self, n, before=None):
        """n steps on the static inputs, the host reading (and checking) step i - 1's record after enqueuing step i.  -> the counts of the last step."""
        counts = None
        first = self.submitted
        for j in range(n):
            t = self.submit(before=before)
            if j >= 1:
                counts = self.counts(t - 1)
        if n >= 1:
            counts = self.counts(first + n - 1)
        return counts
