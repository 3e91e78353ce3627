"""Image-stripe data parallelism: one process per GPU, the BVH replicated, the frame cut into
interleaved row stripes, one framebuffer gather per frame over torch.distributed (RCCL on GPUs, gloo in CPU tests).

Every (pixel, sample) is seeded by its global pixel index and sample index only (draw.cu:162), so any partition gives
the same bytes; stripes are interleaved because scenes are not uniformly expensive (tenthousand.txt's upper half is sky).
"""
import torch
import torch.distributed as dist


class StripePartition:
    """Stripe i (rows [i*stripe_rows, (i+1)*stripe_rows)) belongs to part i % num_parts; a part's buffer holds its
    stripes in increasing order, each row-major (the layout MirtRenderParams describes, include/mirt.h)."""

    def __init__(self, width, height, stripe_rows, num_parts):
        if width <= 0 or height <= 0 or stripe_rows <= 0 or num_parts <= 0:
            raise ValueError("bad partition")
        self.width, self.height, self.stripe_rows, self.num_parts = width, height, stripe_rows, num_parts
        self.num_stripes = (height + stripe_rows - 1) // stripe_rows

    def rows(self, part):
        out = []
        for s in range(part, self.num_stripes, self.num_parts):
            out.extend(range(s * self.stripe_rows, min((s + 1) * self.stripe_rows, self.height)))
        return out

    def num_pixels(self, part):
        return len(self.rows(part)) * self.width

    def max_pixels(self):
        return max(self.num_pixels(p) for p in range(self.num_parts))

    def params(self, part, spp, counters=False):
        from . import api
        return api.render_params(self.width, self.height, spp, self.stripe_rows, self.num_parts, part, counters)


class FrameGatherer:
    """Gathers the per-rank part buffers (uint8 RGBA, padded to the largest part) to rank 0 and re-interleaves them
    into a row-major frame.  Works on whatever device the tensors live on (the collective is the process group's)."""

    def __init__(self, partition, rank, world, device, group=None):
        self.p, self.rank, self.world, self.group = partition, rank, world, group
        self.row_bytes = partition.width * 4
        self.max_bytes = partition.max_pixels() * 4
        self.frame = None
        self.gathered = None
        if rank == 0:
            self.frame = torch.zeros(partition.height * self.row_bytes, dtype=torch.uint8, device=device)
            self.row_index = [torch.tensor(partition.rows(r), dtype=torch.long, device=device) for r in range(world)]
            if world > 1:
                self.gathered = [torch.zeros(self.max_bytes, dtype=torch.uint8, device=device) for _ in range(world)]

    def new_part_buffer(self, device):
        return torch.zeros(self.max_bytes, dtype=torch.uint8, device=device)

    def gather(self, part_buf):
        """part_buf: this rank's compact stripes, padded to max_bytes.  Returns the frame on rank 0, None elsewhere."""
        if self.world > 1:
            if part_buf.is_cuda and dist.get_backend(self.group) == "gloo":
                # rehearsal only (several ranks sharing one GPU, where RCCL refuses duplicate devices): stage through the host
                host = part_buf.cpu()
                got = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(host, got, dst=0, group=self.group)
                if self.rank == 0:
                    for r in range(self.world):
                        self.gathered[r].copy_(got[r])
            else:
                dist.gather(part_buf, self.gathered if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank != 0:
            return None
        fr = self.frame.view(self.p.height, self.row_bytes)
        for r in range(self.world):
            src = self.gathered[r] if self.world > 1 else part_buf
            n = self.row_index[r].numel()
            fr.index_copy_(0, self.row_index[r], src[: n * self.row_bytes].view(n, self.row_bytes))
        return self.frame
