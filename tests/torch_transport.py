"""torch.distributed (gloo) as the transport of the multi-process CPU tests: an adapter with the two shapes the product's host code
knows -- the object collectives of vp8oclenc_amd.ref_shard.ThreadGroup (is_initialized / get_rank / get_world_size / barrier /
all_gather_object / broadcast_object) and rank / world / gather_bytes of vp8oclenc_amd.api.Group.  Test infrastructure: the product
package holds no torch.distributed code (its process group is the library's own, vp8hip_group_*; RCCL inside libvp8hip.so)."""
from __future__ import annotations

import numpy as np


class TorchObjectGroup:
    object_collectives = True

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def is_initialized(self):
        return self.dist.is_initialized()

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def barrier(self):
        self.dist.barrier()

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def broadcast_object(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def gather_bytes(self, data, root: int = 0):
        """api.Group.gather_bytes: every rank's bytes end to end in rank order on `root`: (buffer, counts) there, (None, counts) elsewhere"""
        a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        parts = self.all_gather_object(a.tobytes() if self.rank != root else a.tobytes())
        counts = np.array([len(p) for p in parts], np.uint64)
        if self.rank != root:
            return None, counts
        return np.frombuffer(b"".join(parts), np.uint8), counts
