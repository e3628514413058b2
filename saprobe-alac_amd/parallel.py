"""Multi-GPU: packets shard by contiguous index range, one decoder (one HIP stream) per device, no collective.

The path shards perfectly: no state crosses packets (decoder.go:283,298-300,433-465) and the
PCM slot of packet i is out + i*stride. Two ways to drive it:

* one process per GPU (bench.py under torch.distributed.run): every rank calls `shard_range` and decodes its
  own slice; nothing is exchanged on the data path. `gather_status` exists for callers that want the frame
  counts / status words of the whole batch on every rank (a few bytes per packet, off the hot path).
* one process, several devices (`ShardedDecoder`): hipSetDevice per slice behind the C ABI, one host thread per
  device, as the north star words it.
"""
import threading

import numpy as np


def shard_range(n, world, rank):
    """Packets [lo, hi) of rank `rank`: GPU g of G gets [g*ceil(n/G), min(n, (g+1)*ceil(n/G)))."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def gather_status(frames, status, n_total, world, rank, dist):
    """All ranks get the batch-wide (frames, status) arrays; `dist` is torch.distributed (gloo or nccl)."""
    import torch
    per = (n_total + world - 1) // world
    buf = torch.zeros((2, per), dtype=torch.int64)
    lo, hi = shard_range(n_total, world, rank)
    buf[0, :hi - lo] = torch.from_numpy(np.asarray(frames, dtype=np.int64))
    buf[1, :hi - lo] = torch.from_numpy(np.asarray(status, dtype=np.int64))
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    fr = np.concatenate([p[0].numpy() for p in parts])[:n_total].astype(np.uint32)
    st = np.concatenate([p[1].numpy() for p in parts])[:n_total].astype(np.int32)
    return fr, st


class ShardedDecoder:
    """DecodePackets over several GPUs from one process: a PacketDecoder per device, a host thread per slice."""

    def __init__(self, config, devices, make_decoder=None):
        if make_decoder is None:
            from . import NewPacketDecoder as make_decoder
        self.devices = list(devices)
        self.decoders = [make_decoder(config, d) for d in self.devices]

    def close(self, trim=True):
        """Destroys the per-device decoders; trim: also empty the library's handle pool (a sharded batch decoder's
        workspaces are the size of its shards: nothing a later small decoder would want to inherit)."""
        for d in self.decoders:
            d.close()
        if trim and self.decoders:
            trim_fn = getattr(type(self.decoders[0]), "trim", None)
            if trim_fn is not None:
                trim_fn()

    def decode_batch(self, blob, offsets):
        """Host blob + offsets[n+1] -> (out[n, stride], frames, status), slices decoded concurrently."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        world = len(self.decoders)
        results = [None] * world
        errors = []

        def work(g):
            try:
                lo, hi = shard_range(n, world, g)
                # every slice names its packets in the caller's blob: the host entry uploads only the span its packets
                # cover and checks every descriptor against the blob's length (ALACGPU_ERR_RANGE)
                results[g] = self.decoders[g].decode_batch(blob, offsets[lo:hi + 1])
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(g,)) for g in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        out = np.concatenate([r[0] for r in results if r[0].shape[0]], axis=0) if n else results[0][0]
        frames = np.concatenate([r[1] for r in results])
        status = np.concatenate([r[2] for r in results])
        return out, frames, status
